// nsk_device.h -- device-side building blocks of the Gibbs sweep (gfx950).
//
// Everything here is the per-variable rule of the reference, restated for one GPU lane:
//   eval_factor   numbskull/inference.py:149-413
//   potential     numbskull/inference.py:55-71
//   draw_sample   numbskull/inference.py:36-52
// plus the two generators (Philox4x32-10 for the chromatic scan, MT19937 for the sequential
// validation scan) and the deterministic exp whose algorithm is specified in DESIGN.md so that
// the CPU oracle reproduces it bit for bit.  Compiled with -ffp-contract=off: products and sums
// are rounded separately exactly like the reference's float64 arithmetic; every fused
// multiply-add below is an explicit fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsk {

// ------------------------------------------------------------------------------------------
// factor function ids (inference.py:74-143)
// ------------------------------------------------------------------------------------------
enum : int {
    F_NOOP = -1, F_IMPLY_NATURAL = 0, F_OR = 1, F_AND = 2, F_EQUAL = 3, F_ISTRUE = 4,
    F_LINEAR = 7, F_RATIO = 8, F_LOGICAL = 9, F_AND_CAT = 12, F_IMPLY_MLN = 13, F_OR_CAT = 14,
    F_EQUAL_CAT_CONST = 15, F_IMPLY_NATURAL_CAT = 16, F_IMPLY_MLN_CAT = 17,
    F_DP_GEN_CLASS_PRIOR = 18, F_DP_GEN_LF_PRIOR = 19, F_DP_GEN_LF_PROPENSITY = 20,
    F_DP_GEN_LF_ACCURACY = 21, F_DP_GEN_LF_CLASS_PROPENSITY = 22, F_DP_GEN_DEP_FIXING = 23,
    F_DP_GEN_DEP_REINFORCING = 24, F_DP_GEN_DEP_EXCLUSIVE = 25, F_DP_GEN_DEP_SIMILAR = 26,
    F_UFO = 30
};

// p_info word of a variable: cardinality << 9 | (dataType != 0) << 8 | (uint8) isEvidence
#define NSK_INFO_CARD(i) ((int)((i) >> 9))
#define NSK_INFO_DT1(i) (((i) >> 8) & 1u)
#define NSK_INFO_EV(i) ((int)(int8_t)((i) & 0xffu))
// f_head word of a factor: arity << 8 | (uint8)(factorFunction + 1)
#define NSK_FHEAD_FUNC(h) ((int)((h) & 0xffu) - 1)
#define NSK_FHEAD_ARITY(h) ((int)((h) >> 8))

#define NSK_ZLOCAL 16      // cardinalities up to this keep their running sums in registers/scratch
#define NSK_GRAD_SCALE 4294967296.0   // gradients accumulate as Q31.32 fixed point (order-free)
#define NSK_SMALLW 256      // graphs with at most this many weights accumulate per block in LDS
#define NSK_XCDS 8          // accelerator dies of the MI355X: private copies of the global accumulators

// ------------------------------------------------------------------------------------------
// deterministic exp -- same operation sequence as oracle/nsk_oracle.c:orc_exp_det
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double pow2i(int k) {
    return __longlong_as_double((long long)(k + 1023) << 52);
}

__device__ __forceinline__ double nsk_exp(double x) {
    // same values as the oracle's early returns, written as selects so that no lane branches
    const double INV_LN2 = 1.4426950408889634;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = rint(x * INV_LN2);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    k = k > 1100 ? 1100 : (k < -1100 ? -1100 : k);     // keeps pow2i's bit pattern well-formed
    int k1 = k / 2;
    int k2 = k - k1;
    double res = (p * pow2i(k1)) * pow2i(k2);
    res = (x > 709.782712893384) ? __longlong_as_double(0x7ff0000000000000LL) : res;
    res = (x < -745.1332191019412) ? 0.0 : res;
    return (x != x) ? x : res;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10, counter = (variable id, stream, sweep lo, sweep hi), key = seed
// ------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                            uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // (three-input xor as ONE v_bitop3_b32, truth table 0x96: the rounds are half of the table
        // kernels' vector instructions)
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// The same block with the ten rounds' keys handed in (k0 + r * 0x9E3779B9, k1 + r * 0xBB67AE85): a kernel that evaluates
// a block per trip keeps them in vector registers (philox_round_keys) -- the scalar file is the scarce one there, and the
// compiler re-derives the keys with twenty scalar adds per block otherwise.
struct PhiloxKeys { uint32_t a[10], b[10]; };
__device__ __forceinline__ PhiloxKeys philox_round_keys(uint32_t k0, uint32_t k1) {
    PhiloxKeys pk;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t ka = k0 + (uint32_t)r * 0x9E3779B9u, kb = k1 + (uint32_t)r * 0xBB67AE85u;
        asm volatile("v_mov_b32 %0, %1" : "=v"(pk.a[r]) : "s"(ka));      // (opaque: not re-derived from k0 / k1 later)
        asm volatile("v_mov_b32 %0, %1" : "=v"(pk.b[r]) : "s"(kb));
    }
    return pk;
}
__device__ __forceinline__ u32x4 philox4x32_keyed(const PhiloxKeys &pk, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, pk.a[round], 0x96);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, pk.b[round], 0x96);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    return u32x4{c0, c1, c2, c3};
}

// Generator ids.  A variable's id q is its position in the compiled layout.
// Inference sweeps: ids q and q + 64 with equal q >> 7 -- the same lane of two consecutive tiles --
// share ONE Philox block: counter ((q >> 7) * 64 + (q & 63), 0, sweep), words 0-1 for the lower id
// and 2-3 for the upper one; a kernel that holds both in one lane computes the block once, any other
// computes it per variable.  Learning sweeps need two uniforms per variable and use counter
// (q, stream, sweep) directly.
__device__ __forceinline__ uint32_t inf_block(uint32_t q) { return ((q >> 7) << 6) | (q & 63u); }
__device__ __forceinline__ uint2 inf_words(uint32_t k0, uint32_t k1, uint32_t q, uint32_t s0, uint32_t s1) {
    const u32x4 r = philox4x32(k0, k1, inf_block(q), 0u, s0, s1);
    return ((q >> 6) & 1u) ? uint2{r.z, r.w} : uint2{r.x, r.y};
}
// Positions inside segments with draw tables use the QUAD scheme instead: ids q, q + 64, q + 128, q + 192
// with equal q >> 8 -- the same lane of four consecutive tiles -- share TWO blocks, counter
// ((q >> 8) * 64 + (q & 63), stream, sweep): stream 2 ("A") word (q >> 6) & 3 is the variable's HIGH word a,
// stream 3 ("B") the same word its LOW word b; the 53-bit integer of the draw is (a >> 5) << 26 | b >> 6 as
// ever.  The table kernel compares a >> 5 with the threshold's top 27 bits first -- one block per lane
// decides four updates -- and evaluates block B only on a tie (probability 2^-27 per update); the
// decision is the 53-bit one by construction.  Kernels that need the uniform itself compute both.
__device__ __forceinline__ uint32_t quad_block(uint32_t q) { return ((q >> 8) << 6) | (q & 63u); }
__device__ __forceinline__ uint32_t word_of(const u32x4 &r, uint32_t j) {
    return j == 0u ? r.x : (j == 1u ? r.y : (j == 2u ? r.z : r.w));
}
__device__ __forceinline__ uint2 inf_words_quad(uint32_t k0, uint32_t k1, uint32_t q, uint32_t s0, uint32_t s1) {
    const u32x4 a = philox4x32(k0, k1, quad_block(q), 2u, s0, s1);
    const u32x4 b = philox4x32(k0, k1, quad_block(q), 3u, s0, s1);
    const uint32_t j = (q >> 6) & 3u;
    return uint2{word_of(a, j), word_of(b, j)};
}

// Positions inside WIDE quads (nsk_compile.h seg_wide: one lane samples four consecutive positions) use the same
// two blocks per lane and quad, dealt the other way round (the WIDE scheme): ids 4 i .. 4 i + 3 with equal q >> 2
// share them -- counter ((q >> 8) * 64 + ((q >> 2) & 63), stream, sweep), stream 2 word q & 3 the high word, stream
// 3 the same word the low word.  A lane of the wide path then holds its four draws in its own block.
__device__ __forceinline__ uint32_t wide_block(uint32_t q) { return ((q >> 8) << 6) | ((q >> 2) & 63u); }
__device__ __forceinline__ uint2 inf_words_wide(uint32_t k0, uint32_t k1, uint32_t q, uint32_t s0, uint32_t s1) {
    const u32x4 a = philox4x32(k0, k1, wide_block(q), 2u, s0, s1);
    const u32x4 b = philox4x32(k0, k1, wide_block(q), 3u, s0, s1);
    return uint2{word_of(a, q & 3u), word_of(b, q & 3u)};
}
// ... and a wave that samples such a quad tile by tile (lane l of tile k = offset 64 k + l of the quad) finds the word
// of its position in the block lane 16 k + (l >> 2) computed -- `r` = this lane's block of the quad, (q >> 8, lane)
__device__ __forceinline__ uint32_t wide_word_of_tile(const u32x4 &r, uint32_t k, uint32_t lane) {
    const int src = (int)((16u * k + (lane >> 2)) << 2);
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r.x), w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r.y);
    const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r.z), w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r.w);
    const uint32_t j = lane & 3u;
    return j == 0u ? w0 : (j == 1u ? w1 : (j == 2u ? w2 : w3));
}

// 53-bit uniform in [0,1) from two 32-bit words: the genrand_res53 construction that
// np.random.rand() / random.random() use (inference.py:50, learning.py:90)
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// ------------------------------------------------------------------------------------------
// MT19937 (sequential validation scan only; one lane)
// ------------------------------------------------------------------------------------------
struct MTState { uint32_t mt[624]; int idx; };

__device__ inline uint32_t mt_next(MTState *s) {
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ inline double mt_res53(MTState *s) {
    uint32_t a = mt_next(s);
    uint32_t b = mt_next(s);
    return u53(a, b);
}

// ------------------------------------------------------------------------------------------
// device view of a compiled graph (layout: DESIGN.md "Data layout in HBM")
// ------------------------------------------------------------------------------------------
template <typename VT>
struct DevGraph {
    // per position (owned variables in colour-major order)
    const int32_t *p_vid;       // the caller's variable id (-1: padding position)
    const uint32_t *p_info;     // cardinality / dataType / isEvidence
    const int32_t *p_slot;      // first slot of the variable in slot_off
    const int32_t *p_cnt;       // cstart[vid]: base index into the tally
    const VT *p_init;           // initialValue (evidence value of the evidence chain)
    // Variables are renumbered internally: a sampled variable's internal id IS its position, the
    // variables this handle reads but does not sample (ghosts, isEvidence == 4) follow.  Every id in
    // the arrays below and in the tiles is internal; val / val_evid / v_card are indexed by it, so a
    // class's own stores are contiguous and its neighbour gathers walk the other classes' value
    // ranges in step with the lanes.  p_vid keeps the caller's id for the counter-based generator.
    // inverted index, compacted and laid out in position order
    const int32_t *slot_off;    // [nslot+1] offsets into fidx
    const int32_t *fidx;        // sorted-unique factor ids per (variable, value) slot
    // per factor: one 16-byte record {arity << 8 | function+1, ftv_offset, weightId, 0}
    const uint4 *f_rec;
    const double *f_feat;       // featureValue
    // per edge: one 8-byte record {variable id, dense_equal_to}
    const int2 *m_rec;
    // generic path, one lane per variable: the factor records of every slot copied inline, in list
    // order (DESIGN.md "inline generic stream"): per factor 4 header units of 8 bytes
    // {head, weightId} {ftv_offset, factor id} {featureValue} {members stored, 0} + members {vid, deo}
    const uint2 *gstream;
    const uint32_t *gs_off;     // [nslot] first unit of a slot's records (0 for slots outside this path)
    // per variable id
    const int32_t *v_card;      // cardinality (data-programming "abstain" lookups)
    // weights
    double *w;
    const uint8_t *w_fixed;
    const double *logtab;       // log(k), k = 0..: RATIO's math.log(res) computed on the host
    // state
    VT *val;                    // var_value[0]
    VT *val_evid;               // var_value_evid[0]
    int32_t *cnt;               // tally delta since the last fold into the int64 master copy
    // learning accumulators (per weight), global flavour: NSK_XCDS private copies, one per XCD
    // (copy x at [x * nweight, (x + 1) * nweight)); k_apply_weights adds them up
    long long *G;               // fixed-point gradient sum
    uint32_t *K;                // visits
    uint32_t *T;                // truncating visits (L1)
    // learning accumulators, binned flavour (graphs with <= NSK_SMALLW weights)
    long long *part_G;          // [NSK_LEARN_BINS][nweight]
    uint32_t *part_K, *part_T;
    int32_t nweight;
    int32_t acc_copies;         // copies of G / K / T: NSK_XCDS (one per XCD) or 1
    int32_t bins_xcd;           // the SMALLW bins are dealt to XCDs (8 each): a block adds to its own XCD's bins in that L2
    int32_t packed_grad;        // integer gradients: visit counts ride in the low half of G (GradSink)
    // gradients accumulate as fixed point Q(31+s).(32-s): s = 0 unless the bound on one weight's
    // gradient sum in one class would overflow Q31.32 (nsk_compile.cpp grad_shift)
    long long grad_mul;         // 2^(32-s)
    double grad_inv;            // 2^-(32-s)
    // direct weights (nsk_compile.h w_direct): bit w set = weight w has one factor, hence at most one visit per
    // colour class, and the kernel applies its update at that visit with the class's parameters below
    const uint32_t *w_direct;   // null: none
    double upd_step, upd_reg_param, upd_truncation, upd_cap;
    double upd_a1;              // 1 / (1 + reg_param * step): the L2 factor of a single visit (the same division the
                                //  update launch and the oracle perform, done once on the host)
    int32_t upd_regularization;
    unsigned int *upd_clipped;
    // fast path: inlined adjacency streams (DESIGN.md "fast path") and a position-indexed tally
    const uint4 *adj;           // stream: chunk c of lane i of a tile at adj[off + 64*c + i]
    const uint4 *tiles;         // [nwb] {stream offset, words per lane, tile_hdr offset | PAD, entries}
    const uint32_t *tile_hdr;   // slot programs of uniform tiles, padded to 8 words
    const uint32_t *tile_wrow;  // [nwb] shape tiles: first row of the tile in adj_wt
    double *adj_wt;             // materialised weights of shape tiles: row r, lane i at adj_wt[64 r + i]
    const double *prog_w;       // [2 * |tile_hdr|] per program word: weight * value when the entry
                                //  is satisfied / unsatisfied (0, 0 unless the slot closes an entry);
                                //  refreshed by k_refresh_prog_weights whenever weights change
    uint8_t *cnt_pos;           // [npos] tally delta of binary fast-path variables, by position
                                //        (folded into the int64 master copy every 255 sweeps)
    const uint4 *hub_desc;      // [npos] entry-parallel hubs: {offset into hub_adj, entries, M | card << 8, 0};
    const uint32_t *hub_adj;    //  entries == 0: generic hub walk.  Rows of 64 words, one entry per lane
    uint8_t *sink;              // 1 KiB scratch: where padding lanes store (branch-free epilogues)
    const uint4 *seg_aff;       // implicit adjacency of table segments: slot bases per tile (nsk_compile.h)
    const uint32_t *seg_wide;   // wide quads of table segments: slot bases per quad + exceptions (nsk_compile.h)
    const uint2 *wide_exc;
    const uint4 *ztab;          // draw tables of the uniform programs whose members are all binary:
                                //  entry (program, neighbourhood bits) = {K >> 26, K & (2^26 - 1), sat0 | sat1 << 8, 0}
                                //  (k_refresh_ztab; DESIGN.md "draw tables")
    const uint32_t *bighub_pos; // positions of the hubs a whole workgroup evaluates (hub_desc[..].w = 1)
    const uint4 *ep_desc;       // entry-parallel groups of general tiles (nsk_compile.h ep_desc): one per
    const uint32_t *ep_adj;     //  256 positions; ep_adj: sub-rows of 64 words
    const uint32_t *ep_wrow;    // [groups + 1] first row of a group in ep_wt
    const uint32_t *ep_win, *ep_win_off;   // value windows of the groups (nsk_compile.h ep_win); null: none
    const uint32_t *ep_kstat;   // structural visit counts (nsk_compile.h ep_kstat) or null
    double *ep_wt;              // materialised weights of the groups' entries: row r, lane i at ep_wt[64 r + i]
    const int32_t *iid_of_vid;  // variable id -> internal id (position; ghosts after the positions): the
                                //  literal head lookup of the generic path needs it (uploaded only then)
    int32_t nvar;
    int32_t head_by_vid;
};

// ------------------------------------------------------------------------------------------
// fast path: stream words
//   header  = (function+1) << 27 | nother << 24 | weightId      (0xFFFFFFFF = padding)
//   member  = variable id of a member other than the sampled variable
// The sampled variable is binary and a member of every factor in its stream; all functions are
// symmetric boolean ones, so a factor's value for candidate k follows from three facts about
// the OTHER members: all non-zero? any == 1? all equal (and to what)?
// ------------------------------------------------------------------------------------------
// 16-byte load of streamed-once data (adjacency tiles) with the non-temporal hint, so that it does
// not evict the gathered arrays (values, weights) from the L2
typedef unsigned int nsk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream_load(const uint4 *p) {
    const nsk_u32x4 v = __builtin_nontemporal_load((const nsk_u32x4 *)p);
    return uint4{v.x, v.y, v.z, v.w};
}

#define NSK_PAD_WORD 0xFFFFFFFFu
// Wave-uniform, read-only data (tile descriptors, shared headers, weights) is read through the
// constant address space so that it travels on the scalar path (s_load) instead of occupying 64
// lanes of the vector memory pipeline.
#define NSK_SCALAR __attribute__((address_space(4)))
#define NSK_HDR_FUNC(h) ((int)((h) >> 27) - 1)
#define NSK_HDR_NOTHER(h) ((int)(((h) >> 24) & 7u))
#define NSK_HDR_WID(h) ((int)((h) & 0xFFFFFFu))

struct FactorAcc {
    double w;
    int func, rem, first;
    bool allnz, any1, alleq;
    __device__ __forceinline__ void start(uint32_t hdr, double weight) {
        func = NSK_HDR_FUNC(hdr); rem = NSK_HDR_NOTHER(hdr); w = weight;
        first = -1; allnz = true; any1 = false; alleq = true;
    }
    __device__ __forceinline__ void member(int x) {
        allnz = allnz && (x != 0);
        any1 = any1 || (x == 1);
        if (first < 0) first = x; else alleq = alleq && (x == first);
        rem--;
    }
    // value of the factor with the sampled variable at 0 / at 1 (inference.py:162-200)
    __device__ __forceinline__ void values(double &e0, double &e1) const {
        switch (func) {
        case F_EQUAL:
            e0 = (alleq && (first < 0 || first == 0)) ? 1.0 : -1.0;
            e1 = (alleq && (first < 0 || first == 1)) ? 1.0 : -1.0;
            break;
        case F_AND:
        case F_ISTRUE:
            e0 = -1.0; e1 = allnz ? 1.0 : -1.0;
            break;
        case F_OR:
            e0 = any1 ? 1.0 : -1.0; e1 = 1.0;
            break;
        case F_IMPLY_NATURAL:
            e0 = 0.0; e1 = allnz ? 1.0 : 0.0;
            break;
        default:            // NOOP
            e0 = 0.0; e1 = 0.0;
            break;
        }
    }
};

// blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, not a
// contract): give every XCD one contiguous eighth of the logical blocks so that the value lines
// neighbouring waves share are fetched into one L2 instead of eight.  Launch 8*ceil(n/8) blocks;
// returns -1 for the surplus.
__device__ __forceinline__ int xcd_logical_block(int b, int nblocks) {
    const int per = (nblocks + 7) >> 3;
    const int lb = (b & 7) * per + (b >> 3);
    return lb < nblocks ? lb : -1;
}

// member value at absolute edge index l with the sampled variable hypothetically at `value`
template <typename VT>
__device__ __forceinline__ int member(const int2 *mb, int l, int var_samp, int value,
                                      const VT *val) {
    const int vid = mb[l].x;
    return vid == var_samp ? value : (int)val[vid];
}

// member value and its dense_equal_to in one 8-byte load (categorical functions)
template <typename VT>
__device__ __forceinline__ int member_deo(const int2 *mb, int l, int var_samp, int value,
                                          const VT *val, int &deo) {
    const int2 m = mb[l];
    deo = m.y;
    return m.x == var_samp ? value : (int)val[m.x];
}

// head of IMPLY_MLN / IMPLY_NATURAL_CAT / IMPLY_MLN_CAT: literal reference indexing reads
// var_value[l] with l the ABSOLUTE EDGE INDEX (inference.py:243,277,292); NSK_FLAG_HEAD_BY_VID
// selects the intended fmap[l].vid.  nsk_graph_create has verified l < nvar in literal mode.
template <typename VT>
__device__ __forceinline__ int head_member(const DevGraph<VT> &g, const int2 *mb, int l, int var_samp,
                                           int value, const VT *val, int &deo) {
    const int2 m = mb[l];
    deo = m.y;
    if (m.x == var_samp) return value;
    return (int)val[g.head_by_vid ? m.x : g.iid_of_vid[l]];          // values live at internal ids
}

// eval_factor (inference.py:149-413).  nsk_graph_create rejects unknown function ids and
// out-of-range member positions, so no error path is needed here.
template <typename VT>
__device__ inline double eval_factor(const DevGraph<VT> &g, const uint4 rec, const int2 *mb,
                                     int var_samp, int value, const VT *val) {
    const uint32_t head = rec.x;
    const int fn = NSK_FHEAD_FUNC(head);
    const int s = (int)rec.y;
    const int e = s + NSK_FHEAD_ARITY(head);
    int deo;
    switch (fn) {
    case F_NOOP:
        return 0.0;
    case F_EQUAL: {                                             // 184-192
        int v = member(mb, s, var_samp, value, val);
        for (int l = s + 1; l < e; l++)
            if (v != member(mb, l, var_samp, value, val)) return -1.0;
        return 1.0;
    }
    case F_AND:
    case F_ISTRUE:                                              // 193-200
        for (int l = s; l < e; l++)
            if (member(mb, l, var_samp, value, val) == 0) return -1.0;
        return 1.0;
    case F_OR:                                                  // 177-183
        for (int l = s; l < e; l++)
            if (member(mb, l, var_samp, value, val) == 1) return 1.0;
        return -1.0;
    case F_IMPLY_NATURAL: {                                     // 162-176 (the loop covers the head)
        for (int l = s; l < e; l++)
            if (member(mb, l, var_samp, value, val) == 0) return 0.0;
        return member(mb, e - 1, var_samp, value, val) ? 1.0 : -1.0;
    }
    case F_LINEAR:
    case F_RATIO:
    case F_LOGICAL: {                                           // 201-231
        int hd = member(mb, e - 1, var_samp, value, val);
        int res = 0;
        for (int l = s; l < e - 1; l++) {
            if (member(mb, l, var_samp, value, val) == hd) {
                if (fn == F_LOGICAL) return 1.0;
                res++;
            }
        }
        if (fn == F_LINEAR) return (double)res;
        if (fn == F_RATIO) return g.logtab[res + 1];
        return 0.0;
    }
    case F_IMPLY_MLN: {                                         // 232-246
        for (int l = s; l < e - 1; l++)
            if (member(mb, l, var_samp, value, val) == 0) return 1.0;
        return head_member(g, mb, e - 1, var_samp, value, val, deo) ? 1.0 : 0.0;
    }
    case F_AND_CAT:
    case F_EQUAL_CAT_CONST:                                     // 251-258
        for (int l = s; l < e; l++)
            if (member_deo(mb, l, var_samp, value, val, deo) != deo) return 0.0;
        return 1.0;
    case F_OR_CAT:                                              // 259-265
        for (int l = s; l < e; l++)
            if (member_deo(mb, l, var_samp, value, val, deo) == deo) return 1.0;
        return -1.0;
    case F_IMPLY_NATURAL_CAT: {                                 // 266-280
        for (int l = s; l < e - 1; l++)
            if (member_deo(mb, l, var_samp, value, val, deo) != deo) return 0.0;
        return head_member(g, mb, e - 1, var_samp, value, val, deo) == deo ? 1.0 : -1.0;
    }
    case F_IMPLY_MLN_CAT: {                                     // 281-295
        for (int l = s; l < e - 1; l++)
            if (member_deo(mb, l, var_samp, value, val, deo) != deo) return 1.0;
        return head_member(g, mb, e - 1, var_samp, value, val, deo) == deo ? 1.0 : 0.0;
    }
    case F_DP_GEN_CLASS_PRIOR:                                  // 301-305
        return member(mb, s, var_samp, value, val) == 1 ? 1.0 : -1.0;
    case F_DP_GEN_LF_PRIOR: {                                   // 306-315
        int l0 = member(mb, s, var_samp, value, val);
        return l0 == 2 ? -1.0 : (l0 == 0 ? 0.0 : 1.0);
    }
    case F_DP_GEN_LF_PROPENSITY: {                              // 316-320
        int l0 = member(mb, s, var_samp, value, val);
        return l0 == g.v_card[mb[s].x] - 1 ? 0.0 : 1.0;
    }
    case F_DP_GEN_LF_ACCURACY:
    case F_DP_GEN_LF_CLASS_PROPENSITY: {                        // 321-346
        int y = member(mb, s, var_samp, value, val);
        int l1 = member(mb, s + 1, var_samp, value, val);
        if (l1 == g.v_card[mb[s + 1].x] - 1) return 0.0;
        if (fn == F_DP_GEN_LF_ACCURACY) return y == l1 ? 1.0 : -1.0;
        return y == 1 ? 1.0 : -1.0;
    }
    case F_DP_GEN_DEP_FIXING:
    case F_DP_GEN_DEP_REINFORCING: {                            // 347-380
        int y = member(mb, s, var_samp, value, val);
        int l1 = member(mb, s + 1, var_samp, value, val);
        int l2 = member(mb, s + 2, var_samp, value, val);
        if (l1 == g.v_card[mb[s + 1].x] - 1) return l2 != 1 ? -1.0 : 0.0;
        if (fn == F_DP_GEN_DEP_FIXING) {
            if (l1 == 0 && l2 == 1 && y == 1) return 1.0;
            if (l1 == 1 && l2 == 0 && y == 0) return 1.0;
        } else {
            if (l1 == 0 && l2 == 0 && y == 0) return 1.0;
            if (l1 == 1 && l2 == 1 && y == 1) return 1.0;
        }
        return 0.0;
    }
    case F_DP_GEN_DEP_EXCLUSIVE: {                              // 381-387
        int l1 = member(mb, s, var_samp, value, val);
        int l2 = member(mb, s + 1, var_samp, value, val);
        int abstain = g.v_card[mb[s].x] - 1;
        return (l1 == abstain || l2 == abstain) ? 0.0 : -1.0;
    }
    case F_DP_GEN_DEP_SIMILAR:                                  // 388-393
        return member(mb, s, var_samp, value, val) == member(mb, s + 1, var_samp, value, val)
                   ? 1.0 : 0.0;
    case F_UFO: {                                               // 398-405
        int v = member(mb, s, var_samp, value, val);
        if (v == 0) return 0.0;
        return (double)member(mb, s + v - 1, var_samp, value, val);
    }
    default:
        return 0.0;
    }
}

// potential (inference.py:55-71): sum over the slot's factor list, in list order, each term
// rounded (product) and then added -- no contraction, so the float64 result equals the
// reference's.
template <typename VT>
__device__ inline double potential(const DevGraph<VT> &g, int var_samp, int value, int slot,
                                   const VT *val) {
    double p = 0.0;
    const int b = g.slot_off[slot], e = g.slot_off[slot + 1];
    for (int k = b; k < e; k++) {
        const uint4 rec = g.f_rec[g.fidx[k]];
        const double t = g.w[rec.z] * eval_factor(g, rec, g.m_rec, var_samp, value, val);
        p = p + t;
    }
    return p;
}

// Both candidates of a binary dataType-0 variable share one factor list: walk it once.  The two
// sums are accumulated exactly like two separate potential() calls would (same order, same ops).
template <typename VT>
__device__ inline void potential2(const DevGraph<VT> &g, int var_samp, int slot, const VT *val,
                                  double &p0, double &p1) {
    p0 = 0.0; p1 = 0.0;
    const int b = g.slot_off[slot], e = g.slot_off[slot + 1];
    for (int k = b; k < e; k++) {
        const uint4 rec = g.f_rec[g.fidx[k]];
        const double w = g.w[rec.z];
        const double t0 = w * eval_factor(g, rec, g.m_rec, var_samp, 0, val);
        const double t1 = w * eval_factor(g, rec, g.m_rec, var_samp, 1, val);
        p0 = p0 + t0;
        p1 = p1 + t1;
    }
}

// The same two functions over the inline generic stream: records of the slot are read one after
// the other (header units, then the members), no index chasing.
template <typename VT>
__device__ inline double potential_inl(const DevGraph<VT> &g, int var_samp, int value, int slot,
                                       const VT *val) {
    double p = 0.0;
    const int n = g.slot_off[slot + 1] - g.slot_off[slot];
    const uint2 *r = g.gstream + g.gs_off[slot];
    for (int k = 0; k < n; k++) {
        const uint2 h0 = r[0], h1 = r[1], h3 = r[3];
        const uint4 rec = {h0.x, h1.x, h0.y, 0u};
        const double t = g.w[h0.y] * eval_factor(g, rec, (const int2 *)(r + 4) - (int)h1.x, var_samp, value, val);
        p = p + t;
        r += 4 + h3.x;
    }
    return p;
}

template <typename VT>
__device__ inline void potential2_inl(const DevGraph<VT> &g, int var_samp, int slot, const VT *val,
                                      double &p0, double &p1) {
    p0 = 0.0; p1 = 0.0;
    const int n = g.slot_off[slot + 1] - g.slot_off[slot];
    const uint2 *r = g.gstream + g.gs_off[slot];
    for (int k = 0; k < n; k++) {
        const uint2 h0 = r[0], h1 = r[1], h3 = r[3];
        const uint4 rec = {h0.x, h1.x, h0.y, 0u};
        const int2 *mb = (const int2 *)(r + 4) - (int)h1.x;
        const double w = g.w[h0.y];
        const double t0 = w * eval_factor(g, rec, mb, var_samp, 0, val);
        const double t1 = w * eval_factor(g, rec, mb, var_samp, 1, val);
        p0 = p0 + t0;
        p1 = p1 + t1;
        r += 4 + h3.x;
    }
}

// draw_sample (inference.py:36-52) given the uniform u: Z[k] = running sum of exp(potential),
// z = u * Z[card-1], result = first k with Z[k] >= z (0 if none, like np.argmax of all-False).
template <typename VT, bool INL = false>
__device__ inline int draw_sample(const DevGraph<VT> &g, int var_samp, uint32_t info, int slot0,
                                  const VT *val, double u) {
    const int card = NSK_INFO_CARD(info);
    const int step = NSK_INFO_DT1(info);          // dataType 1: one factor list per value
    if (card == 2) {
        double p0, p1;
        if (step) {
            p0 = (INL ? potential_inl(g, var_samp, 0, slot0, val) : potential(g, var_samp, 0, slot0, val));
            p1 = (INL ? potential_inl(g, var_samp, 1, slot0 + 1, val) : potential(g, var_samp, 1, slot0 + 1, val));
        } else {
            if (INL) potential2_inl(g, var_samp, slot0, val, p0, p1);
            else potential2(g, var_samp, slot0, val, p0, p1);
        }
        const double z0 = nsk_exp(p0);
        const double z1 = z0 + nsk_exp(p1);
        const double z = u * z1;
        return (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    }
    if (card <= NSK_ZLOCAL) {
        double Z[NSK_ZLOCAL];
        double acc = 0.0;
        for (int k = 0; k < card; k++) {
            const double ek = nsk_exp(INL ? potential_inl(g, var_samp, k, slot0 + step * k, val)
                                           : potential(g, var_samp, k, slot0 + step * k, val));
            acc = (k == 0) ? ek : acc + ek;
            Z[k] = acc;
        }
        const double z = u * acc;
        for (int k = 0; k < card; k++)
            if (Z[k] >= z) return k;
        return 0;
    }
    // large domains: two passes, the second recomputes the identical running sums
    double acc = 0.0;
    for (int k = 0; k < card; k++) {
        const double ek = nsk_exp(INL ? potential_inl(g, var_samp, k, slot0 + step * k, val)
                                           : potential(g, var_samp, k, slot0 + step * k, val));
        acc = (k == 0) ? ek : acc + ek;
    }
    const double z = u * acc;
    double run = 0.0;
    for (int k = 0; k < card; k++) {
        const double ek = nsk_exp(INL ? potential_inl(g, var_samp, k, slot0 + step * k, val)
                                           : potential(g, var_samp, k, slot0 + step * k, val));
        run = (k == 0) ? ek : run + ek;
        if (run >= z) return k;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// wave-per-variable flavour (hubs): the 64 lanes evaluate 64 factors of the list at a time, then
// the terms are added one by one IN LIST ORDER (lane 0's term first), so the float64 sum is the
// very same sequence of additions the one-lane potential() performs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_value(double x, int srclane) {      // srclane is wave-uniform
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename VT>
__device__ inline double wave_potential(const DevGraph<VT> &g, int var_samp, int value, int slot,
                                        const VT *val) {
    const int lane = (int)(threadIdx.x & 63);
    const int b = g.slot_off[slot], e = g.slot_off[slot + 1];
    double p = 0.0;
    for (int base = b; base < e; base += 64) {
        double t = 0.0;
        if (base + lane < e) {
            const uint4 rec = g.f_rec[g.fidx[base + lane]];
            t = g.w[rec.z] * eval_factor(g, rec, g.m_rec, var_samp, value, val);
        }
        const int n = min(64, e - base);
        for (int i = 0; i < n; i++) p = p + lane_value(t, i);
    }
    return p;
}

// both candidates of a binary dataType-0 hub in one walk of its list
template <typename VT>
__device__ inline void wave_potential2(const DevGraph<VT> &g, int var_samp, int slot, const VT *val,
                                       double &p0, double &p1) {
    const int lane = (int)(threadIdx.x & 63);
    const int b = g.slot_off[slot], e = g.slot_off[slot + 1];
    p0 = 0.0; p1 = 0.0;
    for (int base = b; base < e; base += 64) {
        double t0 = 0.0, t1 = 0.0;
        if (base + lane < e) {
            const uint4 rec = g.f_rec[g.fidx[base + lane]];
            const double w = g.w[rec.z];
            t0 = w * eval_factor(g, rec, g.m_rec, var_samp, 0, val);
            t1 = w * eval_factor(g, rec, g.m_rec, var_samp, 1, val);
        }
        const int n = min(64, e - base);
        for (int i = 0; i < n; i++) { p0 = p0 + lane_value(t0, i); p1 = p1 + lane_value(t1, i); }
    }
}

// draw_sample with wave-cooperative potentials; every lane returns the same value
template <typename VT>
__device__ inline int wave_draw_sample(const DevGraph<VT> &g, int var_samp, uint32_t info, int slot0,
                                       const VT *val, double u) {
    const int card = NSK_INFO_CARD(info);
    const int step = NSK_INFO_DT1(info);
    if (card == 2 && !step) {
        double p0, p1;
        wave_potential2(g, var_samp, slot0, val, p0, p1);
        const double z0 = nsk_exp(p0);
        const double z1 = z0 + nsk_exp(p1);
        const double z = u * z1;
        return (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    }
    if (card <= 64) {
        // all (candidate, factor) pairs spread over the lanes, 64 pairs per round trip to memory:
        // candidate-major, so that each candidate's terms are met -- and added -- in list order.
        // Lane k keeps candidate k's potential, then its exp and its running sum Z[k].
        const int lane = (int)(threadIdx.x & 63);
        const int B = g.slot_off[slot0];
        const int n0 = g.slot_off[slot0 + 1] - B;                     // dataType 0: the one list
        const int P = step ? g.slot_off[slot0 + card] - B : n0 * card;
        double myp = 0.0;
        for (int base = 0; base < P; base += 64) {
            const int x = base + lane;
            int kk = 0;
            double t = 0.0;
            if (x < P) {
                int at;
                if (step) {                                           // pair x = entry B + x of slot kk
                    at = B + x;
                    for (int k = 1; k < card; k++) kk += (at >= g.slot_off[slot0 + k]) ? 1 : 0;
                } else {
                    kk = x / n0;
                    at = B + (x - kk * n0);
                }
                const uint4 rec = g.f_rec[g.fidx[at]];
                t = g.w[rec.z] * eval_factor(g, rec, g.m_rec, var_samp, kk, val);
            }
            const int n = min(64, P - base);
            for (int i = 0; i < n; i++) {
                const int ki = __builtin_amdgcn_readlane(kk, i);
                const double ti = lane_value(t, i);
                if (lane == ki) myp = myp + ti;
            }
        }
        const double ek = nsk_exp(myp);
        double acc = 0.0, myZ = 0.0;
        for (int k = 0; k < card; k++) {
            const double e = lane_value(ek, k);
            acc = (k == 0) ? e : acc + e;
            if (lane == k) myZ = acc;
        }
        const double z = u * acc;
        const unsigned long long hit = __ballot(lane < card && myZ >= z);
        return hit ? (int)__ffsll((long long)hit) - 1 : 0;
    }
    // larger domains: two passes, the second recomputes the identical running sums
    double acc = 0.0;
    for (int k = 0; k < card; k++) {
        const double ek = nsk_exp(wave_potential(g, var_samp, k, slot0 + step * k, val));
        acc = (k == 0) ? ek : acc + ek;
    }
    const double z = u * acc;
    double run = 0.0;
    for (int k = 0; k < card; k++) {
        const double ek = nsk_exp(wave_potential(g, var_samp, k, slot0 + step * k, val));
        run = (k == 0) ? ek : run + ek;
        if (run >= z) return k;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Where a block sends its (weight, gradient) visits.  Graphs with few weights (every factor of an
// Ising grid shares one or two) would serialise millions of same-address global atomics, so their
// blocks accumulate in LDS tables and add them to one of NSK_LEARN_BINS binned partial sums when
// they finish (k_apply_bins adds the bins up); graphs with many weights use the global
// accumulators -- one private copy per XCD -- directly.
struct GradSink {
    long long *G;       // fixed-point gradient sums (Q31.32: order-independent, hence deterministic)
    uint32_t *K;        // visits
    uint32_t *T;        // truncating visits (L1)
    // packed: every gradient of the graph is an integer (featureValue in {-1,0,1}, no LINEAR / RATIO
    // / UFO), so the 32 fraction bits of G are free and carry the visit count -- one 64-bit atomic
    // per visit instead of two atomics (global accumulators only; nsk_compile.cpp decides)
    bool packed;
    bool local;         // LDS tables or an XCD-private copy: workgroup-scope adds (sink_add)
    // direct weights (DevGraph::w_direct): updated in place at their one visit of the class
    const uint32_t *w_direct;
    double *w;
    double step, reg_param, truncation, cap, grad_inv, a1;
    int regularization;
    unsigned int *clipped;
};

// The accumulators of a sink live in LDS (SMALLW) or in the XCD-private copy of the global tables
// (open_sink): either way nothing outside the issuing XCD touches them during the launch, so the
// adds are workgroup-scope atomics -- they execute in LDS / in the XCD's own L2.  (Agent-scope
// atomics are carried out on the memory side of the fabric because the eight L2s are not coherent
// with each other: the 3.75 million of a 5M-variable LR class cost 130 us of a 300 us launch.)
// `local`: the table is private to this XCD (or in LDS).  Very large weight tables keep ONE global
// copy (eight would cost more to add up after every class than the atomics save) and use
// agent-scope adds.
__device__ __forceinline__ void sink_add(bool local, unsigned long long *p, unsigned long long v) {
    if (local) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sink_add(bool local, uint32_t *p, uint32_t v) {
    if (local) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a^k by binary exponentiation: the operation sequence of oracle powi_det
__device__ __forceinline__ double powi_det(double a, unsigned long long k) {
    double r = 1.0, b = a;
    while (k) {
        if (k & 1) r *= b;
        k >>= 1;
        if (k) b *= b;
    }
    return r;
}

// The weight update of learning.py:110-125 applied to a whole colour class at once
// (DESIGN.md "device-mode learning" gives the closed forms; the oracle restates them).
// cap: a weight visited k times in the class would move by k * step * (mean gradient); when
// k * step exceeds `cap` the class uses step = cap / k for that weight (DESIGN.md "device-mode
// learning": the per-visit rule of the reference has the same fixed point and is stable at any
// k * step because every visit sees the weight the previous one left).  cap <= 0: no clipping.
__device__ __forceinline__ double apply_update(double x, long long G, uint32_t k, uint32_t t, double step,
                                               int regularization, double reg_param, double truncation,
                                               double cap, unsigned int *clipped, double grad_inv) {
    const double Gf = (double)G * grad_inv;
    if (cap > 0.0 && (double)k * step > cap) {
        step = cap / (double)k;
        if (clipped) atomicAdd(clipped, 1u);
    }
    if (regularization == 2) {
        const double a = 1.0 / (1.0 + reg_param * step);
        x = powi_det(a, (unsigned long long)k) * x;
        x = x - step * Gf;
    } else if (regularization == 1) {
        x = x - step * Gf;
        if (t > 0) {
            const double l1 = (reg_param * step * truncation) * (double)t;
            x = (x > 0) ? fmax(0.0, x - l1) : fmin(0.0, x + l1);
        }
    } else {
        x = x - step * Gf;
    }
    return x;
}

// Add one (weight, gradient) visit per participating lane.  Must be called by all 64 lanes of the
// wave (converged); `have` marks participating lanes.  Lanes sharing the leader's weight id are
// reduced in registers first.
// `count`: the lane's visit adds to the visit count K (false: the visit is counted structurally by the
// weight update, nsk_compile.h ep_kstat -- only its gradient is added).
__device__ __forceinline__ void accumulate_gradient(const GradSink &sk, bool have, int wid,
                                                    long long gfix, bool trunc, bool count = true,
                                                    bool have_w = false, double wval = 0.0) {
    // have_w: the caller still holds the weight's value (`wval`: gathered for the potentials, unchanged since)
    if (sk.w_direct) {          // a weight with one factor: this is its only visit of the class -- update it here
        const bool dir = have && ((sk.w_direct[(uint32_t)wid >> 5] >> ((uint32_t)wid & 31u)) & 1u);
        if (dir) {
            double x = have_w ? wval : sk.w[wid];
            if (sk.cap > 0.0 && sk.step > sk.cap) {         // (a step beyond the cap: the general rule, counts the clip)
                x = apply_update(x, gfix, 1u, trunc ? 1u : 0u, sk.step, sk.regularization, sk.reg_param,
                                 sk.truncation, sk.cap, sk.clipped, sk.grad_inv);
            } else {                                        // apply_update with k = 1, its division hoisted (a1)
                const double Gf = (double)gfix * sk.grad_inv;
                if (sk.regularization == 2) {
                    x = sk.a1 * x;                          // powi_det(a, 1) * x = (1.0 * a) * x
                    x = x - sk.step * Gf;
                } else {
                    x = x - sk.step * Gf;
                    if (sk.regularization == 1 && trunc) {
                        const double l1 = (sk.reg_param * sk.step * sk.truncation) * 1.0;
                        x = (x > 0) ? fmax(0.0, x - l1) : fmin(0.0, x + l1);
                    }
                }
            }
            sk.w[wid] = x;
        }
        have = have && !dir;
    }
    const unsigned long long mask = __ballot(have);
    if (mask == 0) return;
    const int leader = __ffsll((long long)mask) - 1;
    const int lw = __shfl(wid, leader, 64);
    const bool same = have && (wid == lw);
    const unsigned long long smask = __ballot(same);
    const int nsame = __popcll(smask);
    if (nsame >= 4) {
        const long long sum = wave_sum_i64(same ? gfix : 0LL);
        const int nt = __popcll(__ballot(same && trunc));
        const int nc = __popcll(__ballot(same && count));
        if ((int)(threadIdx.x & 63) == leader) {
            sink_add(sk.local, (unsigned long long *)&sk.G[lw], (unsigned long long)(sum + (sk.packed ? nc : 0)));
            if (!sk.packed && nc) sink_add(sk.local, &sk.K[lw], (uint32_t)nc);
            if (nt) sink_add(sk.local, &sk.T[lw], (uint32_t)nt);
        }
        have = have && !same;
    }
    if (have) {
        sink_add(sk.local, (unsigned long long *)&sk.G[wid], (unsigned long long)(gfix + ((sk.packed && count) ? 1 : 0)));
        if (!sk.packed && count) sink_add(sk.local, &sk.K[wid], 1u);
        if (trunc) sink_add(sk.local, &sk.T[wid], 1u);
    }
}

}  // namespace nsk
