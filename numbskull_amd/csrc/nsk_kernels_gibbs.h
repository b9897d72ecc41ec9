// nsk_kernels_gibbs.h -- inference sweep kernels (gibbsthread, numbskull/inference.py:10-33):
// the generic CSR kernel, the inlined-adjacency tile kernels and the homogeneous-segment kernel.
#pragma once

#include "nsk_device.h"

namespace nsk {

#define NSK_BLOCK 256
// min waves per SIMD requested for the generic CSR kernels (measured on the 5M mixed LR graph:
// 8 waves + 160 B of scratch beats 4 waves with the running sums in registers by 14 %)
#ifndef NSK_GENERIC_WAVES
#define NSK_GENERIC_WAVES 8
#endif
#ifndef NSK_GENERIC_LEARN_WAVES
#define NSK_GENERIC_LEARN_WAVES 6
#endif
// persistent grids of the learning kernels
#define NSK_LEARN_FAST_BLOCKS 2048
#define NSK_LEARN_LIST_BLOCKS 512
#define NSK_LEARN_GEN_BLOCKS 2048
#define NSK_LEARN_HEAVY_BLOCKS 512
#define NSK_LEARN_GENERAL_BLOCKS 2048
#define NSK_LEARN_SEG_BLOCKS 2048      // per segment launch, at most 4 launches per colour (nsk_compile.h)

// One colour class of one inference sweep: lane <-> variable at position pbegin + global lane id.
// gibbsthread's loop body (inference.py:20-33) for that variable.
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK, NSK_GENERIC_WAVES) void k_gibbs_phase(DevGraph<VT> g, int pbegin, int pend,
                                                           int sample_evidence, int burnin,
                                                           uint32_t k0, uint32_t k1, uint32_t s0,
                                                           uint32_t s1) {
    const int p = pbegin + (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (p >= pend) return;
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info);
    if (!(ev == 0 || sample_evidence)) return;          // inference.py:24 (ev == 4 never gets a position)
    const int v = g.p_vid[p];
    if (v < 0) return;
    const uint2 r = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const int nv = draw_sample<VT, true>(g, p, info, g.p_slot[p], g.val, u53(r.x, r.y));   // inline stream
    g.val[p] = (VT)nv;                                  // values live at the variable's position
    if (!burnin) {                                      // inference.py:29-33
        const int base = g.p_cnt[p];
        if (NSK_INFO_CARD(info) == 2) g.cnt[base] += nv;
        else g.cnt[base + nv] += 1;
    }
}

// Hubs (long factor lists; every generic-path variable of a colour with few of them): one WAVE per
// variable.  The lanes evaluate 64 (candidate, factor) pairs at a time and the terms are added in
// list order, so values are bit-identical to the one-lane kernel; one lane draws and stores.  Runs
// as the first blocks of the categorical general-tile launch (k_gibbs_general<VT, 8>).
template <typename VT>
__device__ __forceinline__ void heavy_update(const DevGraph<VT> &g, int p, int sample_evidence, int burnin,
                                             uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1) {
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info);
    const int v = g.p_vid[p];
    if (v < 0 || !(ev == 0 || sample_evidence)) return;
    const uint2 r = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const int nv = wave_draw_sample(g, p, info, g.p_slot[p], g.val, u53(r.x, r.y));
    if ((threadIdx.x & 63) == 0) {
        g.val[p] = (VT)nv;
        if (!burnin) {
            const int base = g.p_cnt[p];
            if (NSK_INFO_CARD(info) == 2) g.cnt[base] += nv;
            else g.cnt[base + nv] += 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path of one colour class: binary variables whose factors are symmetric boolean functions.
// One wave owns 64 consecutive positions and walks their inlined adjacency tile; every stream word
// is one coalesced 256-byte load for the wave, member words are followed by a 1-byte gather of the
// neighbour's value.  Words are fetched NSK_CHUNK at a time so that the stream loads, then the
// gathers, are all in flight together.  Same float64 operations, in the same order, as
// k_gibbs_phase (potential(): product, then add, in factor-list order).
// ---------------------------------------------------------------------------------------------
#define NSK_CHUNK 8

// Tile with per-lane headers: every lane parses its own word sequence.
template <typename VT>
__device__ __forceinline__ void tile_potentials_dynamic(const DevGraph<VT> &g, const VT *val,
                                                        const uint4 *sp, int len, double &p0,
                                                        double &p1) {
    FactorAcc acc;
    acc.rem = 0; acc.func = F_NOOP; acc.w = 0.0; acc.first = -1;
    acc.allnz = true; acc.any1 = false; acc.alleq = true;
    for (int j0 = 0; j0 < len; j0 += NSK_CHUNK) {
        const uint4 qa = sp[(size_t)(j0 / 4) * 64];
        const uint4 qb = (j0 + 4 < len) ? sp[(size_t)(j0 / 4 + 1) * 64]
                                        : uint4{NSK_PAD_WORD, NSK_PAD_WORD, NSK_PAD_WORD, NSK_PAD_WORD};
        const uint32_t wd[NSK_CHUNK] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        bool ismem[NSK_CHUNK];          // pure ALU on the words just loaded
        int r = acc.rem;
#pragma unroll
        for (int i = 0; i < NSK_CHUNK; i++) {
            ismem[i] = r > 0;
            if (r > 0) r--;
            else if (wd[i] != NSK_PAD_WORD) r = NSK_HDR_NOTHER(wd[i]);
        }
        int xv[NSK_CHUNK];
        double wv[NSK_CHUNK];
#pragma unroll
        for (int i = 0; i < NSK_CHUNK; i++) {
            xv[i] = ismem[i] ? (int)val[wd[i]] : 0;
            wv[i] = (!ismem[i] && wd[i] != NSK_PAD_WORD) ? g.w[NSK_HDR_WID(wd[i])] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NSK_CHUNK; i++) {
            bool done = false;
            if (ismem[i]) {
                acc.member(xv[i]);
                done = acc.rem == 0;
            } else if (wd[i] != NSK_PAD_WORD) {
                acc.start(wd[i], wv[i]);
                done = acc.rem == 0;
            }
            if (done) {
                double e0, e1;
                acc.values(e0, e1);
                const double t0 = acc.w * e0, t1 = acc.w * e1;
                p0 = p0 + t0;
                p1 = p1 + t1;
            }
        }
    }
}

// Uniform tile: all 64 lanes share one slot program (<= 8 member slots).  The program words and
// the per-slot weight terms (prog_w: weight*value for a satisfied / unsatisfied entry, already
// multiplied by k_refresh_prog_weights) arrive by scalar loads, the member ids by one or two
// 16-byte loads per lane, the neighbour values by byte gathers.  The per-slot update is
// straight-line boolean algebra: program flags are wave-uniform, lane facts are lane masks.
// Padding slots (program word 0) and slots that do not close an entry add an exact 0.0.
struct SlotState {
    int first;
    bool allnz, any1, alleq;
};

__device__ __forceinline__ void slot_step(SlotState &st, uint32_t s, double thi, double tlo, int x,
                                          double &p0, double &p1) {
    const bool F = (s >> 27) & 1u, ig = (s >> 29) & 1u;          // uniform
    const uint32_t code = (s >> 24) & 7u;
    const bool nz = ig || (x != 0), one = !ig && (x == 1);
    st.alleq = F || (st.alleq && (x == st.first));
    st.allnz = (F || st.allnz) && nz;
    st.any1 = (!F && st.any1) || one;
    st.first = F ? x : st.first;
    // "satisfied" for candidate 0 / 1 (inference.py:162-200)
    const bool isEq = code == 4u, isAnd = code == 3u || code == 1u, isOr = code == 2u;
    const bool b0 = (isEq && st.alleq && (ig || st.first == 0)) || (isOr && st.any1);
    const bool b1 = (isEq && st.alleq && (ig || st.first == 1)) || (isAnd && st.allnz) || isOr;
    p0 = p0 + (b0 ? thi : tlo);
    p1 = p1 + (b1 ? thi : tlo);
}

// Entries with exactly one other member and one function code for the whole tile (the shape of
// pairwise models such as the Ising grid): no state, two compares per slot.
template <int CODE>
__device__ __forceinline__ void pair_step(double thi, double tlo, int x, double &p0, double &p1) {
    bool b0, b1;
    if (CODE == 4) { b0 = x == 0; b1 = x == 1; }            // EQUAL
    else if (CODE == 2) { b0 = x == 1; b1 = true; }          // OR
    else { b0 = false; b1 = x != 0; }                        // AND / ISTRUE / IMPLY_NATURAL
    p0 = p0 + (b0 ? thi : tlo);
    p1 = p1 + (b1 ? thi : tlo);
}

template <typename VT, int KIND>
__device__ __forceinline__ void tile_potentials_uniform(const DevGraph<VT> &g, const VT *val,
                                                        const uint4 *sp, int len, uint32_t prog,
                                                        double &p0, double &p1) {
    const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    const NSK_SCALAR double *tw = (const NSK_SCALAR double *)(g.prog_w + 2 * (size_t)prog);
    if (len <= 0) return;
    SlotState st = {0, true, false, true};
    const uint4 qa = sp[0];
    uint32_t sl[4];
    double th[4], tl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { sl[i] = KIND ? 0u : pp[i]; th[i] = tw[2 * i]; tl[i] = tw[2 * i + 1]; }
    const int x0 = (int)val[qa.x], x1 = (int)val[qa.y], x2 = (int)val[qa.z], x3 = (int)val[qa.w];
    if (KIND) {
        pair_step<KIND>(th[0], tl[0], x0, p0, p1);
        pair_step<KIND>(th[1], tl[1], x1, p0, p1);
        pair_step<KIND>(th[2], tl[2], x2, p0, p1);
        pair_step<KIND>(th[3], tl[3], x3, p0, p1);
    } else {
        slot_step(st, sl[0], th[0], tl[0], x0, p0, p1);
        slot_step(st, sl[1], th[1], tl[1], x1, p0, p1);
        slot_step(st, sl[2], th[2], tl[2], x2, p0, p1);
        slot_step(st, sl[3], th[3], tl[3], x3, p0, p1);
    }
    if (len > 4) {
        const uint4 qb = sp[64];
#pragma unroll
        for (int i = 0; i < 4; i++) { sl[i] = KIND ? 0u : pp[4 + i]; th[i] = tw[8 + 2 * i]; tl[i] = tw[9 + 2 * i]; }
        const int x4 = (int)val[qb.x], x5 = (int)val[qb.y], x6 = (int)val[qb.z], x7 = (int)val[qb.w];
        if (KIND) {
            pair_step<KIND>(th[0], tl[0], x4, p0, p1);
            pair_step<KIND>(th[1], tl[1], x5, p0, p1);
            pair_step<KIND>(th[2], tl[2], x6, p0, p1);
            pair_step<KIND>(th[3], tl[3], x7, p0, p1);
        } else {
            slot_step(st, sl[0], th[0], tl[0], x4, p0, p1);
            slot_step(st, sl[1], th[1], tl[1], x5, p0, p1);
            slot_step(st, sl[2], th[2], tl[2], x6, p0, p1);
            slot_step(st, sl[3], th[3], tl[3], x7, p0, p1);
        }
    }
}

// prog_w[2i], prog_w[2i+1] = weight * (value when satisfied, value when not) of program word i, or
// (0, 0) when the slot does not close an entry.  The products are the reference's own
// `weight * eval_factor` (inference.py:68-70), so adding them reproduces potential() exactly.
static __global__ __launch_bounds__(NSK_BLOCK) void k_refresh_prog_weights(const uint32_t *prog, const double *w,
                                                                    double *prog_w, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= n) return;
    const uint32_t s = prog[i];
    if (s >> 31) return;                                   // role word of a shape tile, not a slot
    const uint32_t code = (s >> 24) & 7u;
    const bool last = (s >> 28) & 1u;
    const double hi = code == 0u ? 0.0 : 1.0;
    const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
    const double wt = w[s & 0xFFFFFFu];
    prog_w[2 * i] = last ? wt * hi : 0.0;
    prog_w[2 * i + 1] = last ? wt * lo : 0.0;
}

#define NSK_SHAPE_NULL 0xFFFFFFFFu      // member slot of a shape tile that the lane's entry lacks (nsk_compile.cpp)
// Shape tile: every lane has its own factor functions and weights (per-lane header words in the
// stream) but all 64 lanes share the word layout -- which words are headers, which are members,
// where entries start and end -- so the walk is driven by a scalar role program and only the
// per-lane facts (function code, weight, member values) are vector work.  This is the shape of
// graphs whose factors carry individual weights.
template <typename VT>
__device__ __forceinline__ void tile_potentials_shape(const DevGraph<VT> &g, const VT *val,
                                                      const uint4 *sp, int len, uint32_t prog,
                                                      uint32_t wrow, double &p0, double &p1) {
    const NSK_SCALAR uint32_t *rp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    const double *wt = g.adj_wt + (size_t)wrow * 64 + (threadIdx.x & 63);   // this lane's weight column
    uint32_t code = 0;                   // per lane: 0 NOOP 1 IMPLY_NATURAL 2 OR 3 AND/ISTRUE 4 EQUAL
    double w = 0.0;
    int first = 0, entry = 0;            // entry: wave-uniform
    bool allnz = true, any1 = false, alleq = true;
    auto finish = [&](bool nomember) {
        const bool isEq = code == 4u, isAnd = code == 3u || code == 1u, isOr = code == 2u;
        const bool b0 = (isEq && alleq && (nomember || first == 0)) || (isOr && any1);
        const bool b1 = (isEq && alleq && (nomember || first == 1)) || (isAnd && allnz) || isOr;
        const double hi = code == 0u ? 0.0 : 1.0;
        const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
        const double t0 = w * (b0 ? hi : lo), t1 = w * (b1 ? hi : lo);
        p0 = p0 + t0;
        p1 = p1 + t1;
    };
    for (int c = 0; c * 4 < len; c++) {                                  // <= 4 chunks, scalar loop
        const uint4 q = sp[(size_t)c * 64];
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        uint32_t role[4];
#pragma unroll
        for (int i = 0; i < 4; i++) role[i] = rp[4 * c + i] & 0x1Fu;     // scalar; 0 = padding word
        double wv[4];
        int xv[4];
        int e2 = entry;
#pragma unroll
        for (int i = 0; i < 4; i++) {                                    // one load per word
            wv[i] = 0.0; xv[i] = 0;
            if (role[i] & 1u) wv[i] = wt[(size_t)(e2++) * 64];           // coalesced: materialised weight
            else if ((role[i] & 16u) && wd[i] != NSK_SHAPE_NULL) xv[i] = (int)val[wd[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (role[i] & 1u) {                                          // header: open an entry
                code = (0x343210u >> (4u * (wd[i] >> 27))) & 0xFu;       // function+1 in 0..5 -> code
                w = wv[i];
                entry++;
                first = 0; allnz = true; any1 = false; alleq = true;
                if (role[i] & 8u) finish(true);
            } else if (role[i] & 16u) {                                  // member
                // (a slot this lane's entry does not have -- never its first -- repeats the first member's value:
                //  neutral for "all equal", "all non-zero" and "any is 1")
                const int x = wd[i] == NSK_SHAPE_NULL ? first : xv[i];
                const bool F = (role[i] & 2u) != 0;
                alleq = F || (alleq && (x == first));
                allnz = (F || allnz) && (x != 0);
                any1 = (!F && any1) || (x == 1);
                first = F ? x : first;
                if (role[i] & 4u) finish(false);
            }
        }
    }
}

// adj_wt row e of a shape tile <- the weights its lanes' e-th headers name (run whenever weights
// may have changed; the sweeps then read weights as coalesced rows instead of 64 random sectors)
static __global__ __launch_bounds__(NSK_BLOCK) void k_refresh_shape_weights(const uint4 *tiles, const uint4 *adj,
                                                                     const uint32_t *tile_hdr,
                                                                     const uint32_t *tile_wrow, const double *w,
                                                                     double *adj_wt, int ntiles) {
    const int lane = (int)(threadIdx.x & 63);
    const int t = (int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
    if (t >= ntiles) return;
    const uint4 td = tiles[t];
    const uint32_t kind = (td.w >> 8) & 7u;
    if (td.z == NSK_PAD_WORD || !(kind == 7u || (kind == 6u && ((td.w >> 19) & 1u)))) return;
    const uint32_t mask = kind == 7u ? 0xFFFFFFu : 0xFFFFFFFFu;     // general tiles: the word is the id
    const int len = (int)(td.w & 0xFFu);
    const uint32_t *rp = tile_hdr + td.z;
    double *wt = adj_wt + (size_t)tile_wrow[t] * 64 + lane;
    int entry = 0;
    for (int c = 0; c * 4 < len; c++) {
        const uint4 q = adj[td.x + (size_t)c * 64 + lane];
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        for (int i = 0; i < 4; i++)
            if (rp[4 * c + i] & 1u) wt[(size_t)(entry++) * 64] = w[wd[i] & mask];
    }
}

// ---------------------------------------------------------------------------------------------
// General tiles (kind 6): variables of cardinality <= 8 and either dataType whose factors are the
// symmetric boolean functions, IMPLY_MLN or the *_CAT functions (nsk_compile.cpp "general tiles").
// The 64 lanes share a layout of E entries x (weight word, descriptor word, M member slots); code,
// weight, members and the candidate value an entry belongs to are per lane.  With the other
// members read, every entry is  value(candidate c) = (c == cstar) ? A : B  with A, B in {-1, 0, 1}:
// the restatement of eval_factor (inference.py:162-200, 232-295) as facts about the other members.
// ---------------------------------------------------------------------------------------------
#ifndef NSK_GEN_NULL
#define NSK_GEN_NULL 0x7FFFFFFu
#endif
#define NSK_GEN_GROUP 2
#ifndef NSK_GEN_AHEAD_MAX
#define NSK_GEN_AHEAD_MAX 3
#endif
struct GenChain {
    bool allnz, prevall, any1, alleq, lastnz;
    int first;
    __device__ __forceinline__ void open() {
        allnz = true; prevall = true; any1 = false; alleq = true; lastnz = false; first = 0;
    }
    // one member slot; `cat`: the function compares with dense_equal_to instead of 0 / 1
    __device__ __forceinline__ void member(bool F, bool cat, uint32_t word, int x) {
        const bool live = (word & NSK_GEN_NULL) != NSK_GEN_NULL;      // not an empty slot
        const int deo = (int)(word >> 27);
        const bool nz = cat ? (x == deo) : (x != 0);
        const bool one = cat ? (x == deo) : (x == 1);
        alleq = live ? (F || (alleq && x == first)) : alleq;          // selects, no divergent branch
        first = (live && F) ? x : first;
        prevall = live ? allnz : prevall;
        allnz = live ? (allnz && nz) : allnz;
        any1 = live ? (any1 || one) : any1;
        lastnz = live ? nz : lastnz;
    }
    // (c == cstar) ? A : B for the entry described by descriptor word d1: one byte of the table below
    __device__ __forceinline__ void close(uint32_t d1, const uint8_t *lut, int &cstar, int &A, int &B) const {
        const uint32_t nomember = ((d1 >> 4) & 7u) == 0u ? 1u : 0u;
        // own role (bits 7-8): 1 body, 2 head, 3 body AND head of a positional function -- then every
        // other member is a body member (body test = allnz) and the head test is the constant in bit 18
        // (nsk_compile.cpp general_words)
        const bool both = ((d1 >> 7) & 3u) == 3u;
        const bool pv = both ? allnz : prevall, ln = both ? ((d1 >> 18) & 1u) != 0u : lastnz;
        const uint32_t idx = (d1 & 15u) | (((d1 >> 7) & 1u) << 4) | (nomember << 5) |
                             ((allnz ? 1u : 0u) << 6) | ((pv ? 1u : 0u) << 7) | ((any1 ? 1u : 0u) << 8) |
                             ((alleq ? 1u : 0u) << 9) | ((ln ? 1u : 0u) << 10);
        const uint32_t e = lut[idx];
        A = (int)(e & 3u) - 1;
        B = (int)((e >> 2) & 3u) - 1;
        const uint32_t sel = e >> 4;                    // cstar: 0, 1, first member's value, own dense_equal_to
        const int sdeo = (int)((d1 >> 9) & 31u);
        cstar = sel == 0u ? 0 : (sel == 1u ? 1 : (sel == 2u ? first : sdeo));
    }
};

// The restatement of eval_factor (inference.py:162-200, 232-295) as facts about the other members,
// tabulated: index = code | own role is body << 4 | no other member << 5 | allnz << 6 | prevall << 7 |
// any1 << 8 | alleq << 9 | lastnz << 10; entry = (A + 1) | (B + 1) << 2 | cstar selector << 4.
// (A table instead of branches: the function code is per lane, and the 64-fold unrolled walk with
// ten-way divergent branches per entry did not fit the instruction cache.)
struct GenLut { uint8_t t[2048]; };
constexpr uint8_t gen_lut_entry(uint32_t idx) {
    const uint32_t code = idx & 15u;
    const bool role1 = (idx >> 4) & 1u, nomember = (idx >> 5) & 1u, allnz = (idx >> 6) & 1u,
               prevall = (idx >> 7) & 1u, any1 = (idx >> 8) & 1u, alleq = (idx >> 9) & 1u,
               lastnz = (idx >> 10) & 1u;
    const bool body = role1 ? prevall : allnz;          // body members all true / matching
    const bool hd = lastnz;                             // own role body: the head is the last member
    int sel = 0, A = 0, B = 0;
    if (code == 1u) { B = allnz ? 1 : 0; }                                            // IMPLY_NATURAL
    else if (code == 2u) { sel = 1; A = 1; B = any1 ? 1 : -1; }                       // OR
    else if (code == 3u) { A = -1; B = allnz ? 1 : -1; }                              // AND / ISTRUE
    else if (code == 4u) { sel = 2; A = (nomember || alleq) ? 1 : -1; B = nomember ? 1 : -1; }   // EQUAL
    else if (code == 5u) {                                                            // IMPLY_MLN
        if (role1) { A = 1; B = !body ? 1 : (hd ? 1 : 0); }
        else { A = !allnz ? 1 : 0; B = 1; }
    } else if (code == 6u) { sel = 3; A = allnz ? 1 : 0; }                            // AND_CAT / EQUAL_CAT_CONST
    else if (code == 7u) { sel = 3; A = 1; B = any1 ? 1 : -1; }                       // OR_CAT
    else if (code == 8u) {                                                            // IMPLY_NATURAL_CAT
        sel = 3;
        if (role1) { A = body ? (hd ? 1 : -1) : 0; }
        else { A = allnz ? 1 : 0; B = allnz ? -1 : 0; }
    } else if (code == 9u) {                                                          // IMPLY_MLN_CAT
        sel = 3;
        if (role1) { A = !body ? 1 : (hd ? 1 : 0); B = 1; }
        else { A = 1; B = allnz ? 0 : 1; }
    } else if (code == 11u) { A = 1; B = 1; }           // constant 1 (code 10: constant 0) -- a variable
    return (uint8_t)((A + 1) | ((B + 1) << 2) | (sel << 4));   // whose own edges in the factor disagree
}
constexpr GenLut make_gen_lut() {
    GenLut l{};
    for (uint32_t i = 0; i < 2048; i++) l.t[i] = gen_lut_entry(i);
    return l;
}
static __constant__ GenLut k_gen_lut = make_gen_lut();

// Learning: is the entry with descriptor d1 among the factors sample_and_sgd visits for this variable
// (learning.py:76-95: the sorted-unique union of the lists of the evidence and the proposal value)?
// dataType 0: always (one list).  dataType 1: when its list is selected -- except that a factor the
// variable's own edges put into TWO lists is visited through the smaller value's entry when both are
// selected (descriptor bit 19: has such a partner, bits 20-22: the partner's value).
__device__ __forceinline__ bool entry_visited(uint32_t d1, int evidence, int proposal) {
    const int ks = (int)((d1 >> 14) & 15u);
    const int partner = (int)((d1 >> 20) & 7u);
    const bool dup = ((d1 >> 19) & 1u) != 0u && (partner == evidence || partner == proposal);
    return ks == 15 || ((ks == evidence || ks == proposal) && !dup);
}

// block-wide copy of the table into LDS; every thread of the block must call it
__device__ __forceinline__ void load_gen_lut(uint8_t *lds) {
    for (int i = (int)threadIdx.x; i < 512; i += NSK_BLOCK)
        ((uint32_t *)lds)[i] = ((const uint32_t *)k_gen_lut.t)[i];
    __syncthreads();
}

// Walk a general tile over one (TWO = false) or two value arrays; on_entry(weight id, weight,
// descriptor, chain a, chain b) runs at every entry end, in list order -- wave-uniform control flow,
// so it may use wave collectives.
// WMODE: where an entry's weight comes from -- 0 gathered from g.w, 1 the tile's materialised weight
// rows (wt = this lane's column), 2 not needed (0.0).
// The tile's layout is E entries of (weight word, descriptor word, M member slots) with E and M
// shared by the 64 lanes (nsk_compile.cpp), so the walk is specialised on M and straight-line per
// entry: a "super-group" of EG entries covers a whole number CG of 16-byte chunks (E is padded to a
// multiple of EG by the compiler).  Per super-group: the chunks arrive (the next super-group's are
// requested first), then every weight and member gather is issued, then the entries are closed.
// (The first general kernels decoded a per-word role program with scalar branches: 53 VALU + 38
// SALU instructions per stream word, which made the launch instruction-issue bound.)
template <typename VT, bool TWO, int WMODE, bool NT, int M, typename FN>
__device__ __forceinline__ void general_walk_m(const DevGraph<VT> &g, const VT *va, const VT *vb,
                                               const uint4 *sp, int E, const double *wt, FN &&on_entry) {
    constexpr int W = 2 + M;
    constexpr int EG = (W % 4 == 0) ? 1 : (W % 2 == 0) ? 2 : 4;      // entries per super-group
    constexpr int CG = W * EG / 4;                                    // 16-byte chunks per super-group
    auto load_sg = [&](int e0, uint4 (&q)[CG]) {
#pragma unroll
        for (int j = 0; j < CG; j++) {
            const uint4 *at = sp + (size_t)((e0 / EG) * CG + j) * 64;
            q[j] = NT ? stream_load(at) : *at;                        // NT: the tile is read once (inference)
        }
    };
    // (the next super-group is requested ahead only while that costs few registers: at CG >= 5 the
    // second buffer would push the kernel from 4 to 3 waves per SIMD)
    constexpr bool AHEAD = CG <= NSK_GEN_AHEAD_MAX;
    uint4 qn[CG];
    if (AHEAD && E > 0) load_sg(0, qn);
    for (int e0 = 0; e0 < E; e0 += EG) {
        if (!AHEAD) load_sg(e0, qn);
        uint32_t wd[W * EG];
#pragma unroll
        for (int j = 0; j < CG; j++) {
            wd[4 * j] = qn[j].x; wd[4 * j + 1] = qn[j].y; wd[4 * j + 2] = qn[j].z; wd[4 * j + 3] = qn[j].w;
        }
        if (AHEAD && e0 + EG < E) load_sg(e0 + EG, qn);
        double wv[EG];
        int xa[EG * (M > 0 ? M : 1)], xb[EG * (M > 0 ? M : 1)];
#pragma unroll
        for (int j = 0; j < EG; j++) {
            wv[j] = WMODE == 0 ? g.w[wd[j * W]] : (WMODE == 1 ? wt[(size_t)(e0 + j) * 64] : 0.0);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t id = wd[j * W + 2 + m] & NSK_GEN_NULL;
                const uint32_t at = id == NSK_GEN_NULL ? 0u : id;
                xa[j * M + m] = (int)va[at];
                xb[j * M + m] = TWO ? (int)vb[at] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < EG; j++) {
            const uint32_t d1 = wd[j * W + 1];
            const bool cat = (d1 & 15u) >= 6u;
            GenChain a, b;
            a.open();
            if (TWO) b.open();
#pragma unroll
            for (int m = 0; m < M; m++) {
                a.member(m == 0, cat, wd[j * W + 2 + m], xa[j * M + m]);
                if (TWO) b.member(m == 0, cat, wd[j * W + 2 + m], xb[j * M + m]);
            }
            on_entry(wd[j * W], wv[j], d1, a, b);
        }
    }
}

// Entries with more than 3 other members (factors of arity >= 5 over small-domain variables) are
// rare; their tiles take the role-program walk: one scalar role word per stream word (tile_hdr)
// drives the same chain updates.  Keeping them out of the specialised walk keeps its register
// count at the M <= 3 instantiations'.
template <typename VT, bool TWO, int WMODE, bool NT, typename FN>
__device__ __forceinline__ void general_walk_roles(const DevGraph<VT> &g, const VT *va, const VT *vb,
                                             const uint4 *sp, int len, uint32_t prog, const double *wt,
                                             FN &&on_entry) {
    const NSK_SCALAR uint32_t *rp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    if (len <= 0) return;
    GenChain a, b;
    a.open(); b.open();
    uint32_t wid = 0, d1 = 0;
    double w = 0.0;
    int entry = 0;                                 // wave-uniform
    // NSK_GEN_GROUP chunks (16-byte loads) at a time: their stream loads, then all their gathers,
    // are in flight together -- the walk of a long tile is a chain of dependent memory round trips.
    // The next group's stream loads are issued before this group's gathers (double buffer), so the
    // chain is one round trip per group instead of two.
    auto load_group = [&](int c0, uint4 (&q)[NSK_GEN_GROUP]) {
#pragma unroll
        for (int j = 0; j < NSK_GEN_GROUP; j++)
            q[j] = ((c0 + j) * 4 >= len) ? uint4{0u, 0u, 0u, 0u}                                // uniform guard
                   : (NT ? stream_load(sp + (size_t)(c0 + j) * 64)   // NT: the tile is read once (inference)
                         : sp[(size_t)(c0 + j) * 64]);
    };
    uint4 qn[NSK_GEN_GROUP];
    load_group(0, qn);
    for (int c0 = 0; c0 * 4 < len; c0 += NSK_GEN_GROUP) {
        uint4 q[NSK_GEN_GROUP];
#pragma unroll
        for (int j = 0; j < NSK_GEN_GROUP; j++) q[j] = qn[j];
        if ((c0 + NSK_GEN_GROUP) * 4 < len) load_group(c0 + NSK_GEN_GROUP, qn);
        uint32_t wd[4 * NSK_GEN_GROUP], role[4 * NSK_GEN_GROUP];
#pragma unroll
        for (int j = 0; j < NSK_GEN_GROUP; j++) {
            wd[4 * j] = q[j].x; wd[4 * j + 1] = q[j].y; wd[4 * j + 2] = q[j].z; wd[4 * j + 3] = q[j].w;
#pragma unroll
            for (int i = 0; i < 4; i++)           // scalar; 0 = padding word or beyond the tile
                role[4 * j + i] = ((c0 + j) * 4 + i < len) ? (rp[(c0 + j) * 4 + i] & 0x3Fu) : 0u;
        }
        int xa[4 * NSK_GEN_GROUP], xb[4 * NSK_GEN_GROUP];
        double wv[NSK_GEN_GROUP * 2];              // at most one weight word per two words
        int nwv = 0;
#pragma unroll
        for (int i = 0; i < 4 * NSK_GEN_GROUP; i++) {
            xa[i] = 0; xb[i] = 0;
            if ((role[i] & 1u) && WMODE != 2) {
                const double x = WMODE == 1 ? wt[(size_t)(entry++) * 64] : g.w[wd[i]];
#pragma unroll
                for (int j = 0; j < NSK_GEN_GROUP * 2; j++) if (j == nwv) wv[j] = x;
                nwv++;
            }
            if (role[i] & 16u) {
                const uint32_t id = wd[i] & NSK_GEN_NULL;
                const uint32_t at = id == NSK_GEN_NULL ? 0u : id;
                xa[i] = (int)va[at];
                if (TWO) xb[i] = (int)vb[at];
            }
        }
        int iwv = 0;
#pragma unroll
        for (int i = 0; i < 4 * NSK_GEN_GROUP; i++) {
            if (role[i] & 1u) {
                wid = wd[i];
                if (WMODE != 2) {
#pragma unroll
                    for (int j = 0; j < NSK_GEN_GROUP * 2; j++) if (j == iwv) w = wv[j];
                    iwv++;
                }
            } else if (role[i] & 32u) {
                d1 = wd[i];
                a.open();
                if (TWO) b.open();
                if (role[i] & 8u) on_entry(wid, w, d1, a, b);
            } else if (role[i] & 16u) {
                const bool F = (role[i] & 2u) != 0, cat = (d1 & 15u) >= 6u;
                a.member(F, cat, wd[i], xa[i]);
                if (TWO) b.member(F, cat, wd[i], xb[i]);
                if (role[i] & 4u) on_entry(wid, w, d1, a, b);
            }
        }
    }
}


// `len` = E * (2 + M) words, M = member slots per entry (tile descriptor bits 16..18)
template <typename VT, bool TWO, int WMODE, bool NT, typename FN>
__device__ __forceinline__ void general_walk(const DevGraph<VT> &g, const VT *va, const VT *vb,
                                             const uint4 *sp, int len, int M, uint32_t prog, const double *wt,
                                             FN &&on_entry) {
    if (len <= 0) return;
    const int E = len / (2 + M);
    switch (M) {                                                       // wave-uniform
    case 0: general_walk_m<VT, TWO, WMODE, NT, 0>(g, va, vb, sp, E, wt, on_entry); break;
    case 1: general_walk_m<VT, TWO, WMODE, NT, 1>(g, va, vb, sp, E, wt, on_entry); break;
    case 2: general_walk_m<VT, TWO, WMODE, NT, 2>(g, va, vb, sp, E, wt, on_entry); break;
    case 3: general_walk_m<VT, TWO, WMODE, NT, 3>(g, va, vb, sp, E, wt, on_entry); break;
    default: general_walk_roles<VT, TWO, WMODE, NT>(g, va, vb, sp, len, prog, wt, on_entry); break;
    }
}

// The (weight id, descriptor) words of a general tile's entries, in list order, without member
// gathers: fn(entry index, weight id, descriptor).  M <= 3 only (the layouts of general_walk_m).
template <int M, typename FN>
__device__ __forceinline__ void general_walk_ids_m(const uint4 *sp, int E, FN &&fn) {
    constexpr int W = 2 + M;
    constexpr int EG = (W % 4 == 0) ? 1 : (W % 2 == 0) ? 2 : 4;
    constexpr int CG = W * EG / 4;
    for (int e0 = 0; e0 < E; e0 += EG) {
        uint32_t wd[W * EG];
#pragma unroll
        for (int j = 0; j < CG; j++) {
            const uint4 q = sp[(size_t)((e0 / EG) * CG + j) * 64];
            wd[4 * j] = q.x; wd[4 * j + 1] = q.y; wd[4 * j + 2] = q.z; wd[4 * j + 3] = q.w;
        }
#pragma unroll
        for (int j = 0; j < EG; j++) fn(e0 + j, wd[j * W], wd[j * W + 1]);
    }
}
template <typename FN>
__device__ __forceinline__ void general_walk_ids(const uint4 *sp, int len, int M, FN &&fn) {
    const int E = len / (2 + M);
    switch (M) {                                                       // wave-uniform
    case 0: general_walk_ids_m<0>(sp, E, fn); break;
    case 1: general_walk_ids_m<1>(sp, E, fn); break;
    case 2: general_walk_ids_m<2>(sp, E, fn); break;
    default: general_walk_ids_m<3>(sp, E, fn); break;
    }
}

// potentials of the candidates 0..MAXC-1 of one general tile: p[c] accumulates, in list order, the
// rounded products weight * value exactly like potential() (inference.py:55-71).  MAXC = 2 serves
// the tiles whose lanes are all binary (the compiler sorts those behind the categorical ones).
template <int MAXC>
struct GenPot {
    double p[MAXC];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < MAXC; c++) p[c] = 0.0;
    }
    __device__ __forceinline__ void add(int maxcard, uint32_t d1, double w, int cstar, int A, int B) {
        add_ks(maxcard, (int)((d1 >> 14) & 15u), w, cstar, A, B);
    }
    // ks: candidate owning the entry; 15 = all (dataType 0); 14 = none (padding)
    __device__ __forceinline__ void add_ks(int maxcard, int ks, double w, int cstar, int A, int B) {
        const double tA = w * (double)A, tB = w * (double)B;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c >= 2 && c >= maxcard) break;        // wave-uniform
            const bool on = ks == 15 || ks == c;
            const double t = on ? (c == cstar ? tA : tB) : 0.0;   // +0.0 leaves the sum unchanged
            p[c] = p[c] + t;
        }
    }
    // draw_sample (inference.py:36-52): running sums of exp, first candidate with Z[k] >= u * Z[card-1]
    __device__ __forceinline__ int draw(int maxcard, int card, double u) const {
        double Z[MAXC];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            Z[k] = 0.0;
            if (k >= 2 && k >= maxcard) break;
            const double ek = nsk_exp(p[k]);
            const double next = (k == 0) ? ek : acc + ek;
            acc = k < card ? next : acc;
            Z[k] = acc;
        }
        const double z = u * acc;
        int nv = 0;
#pragma unroll
        for (int k = MAXC - 1; k >= 0; k--) {
            if (k >= 2 && k >= maxcard) continue;
            if (k < card && Z[k] >= z) nv = k;
        }
        return nv;
    }
};

// ---------------------------------------------------------------------------------------------
// Entry-parallel hubs (nsk_compile.cpp "entry-parallel hub streams"): one wave per hub, one LANE
// per list entry.  Every lane evaluates its entry exactly like a general tile does (chain facts,
// gen_lut_entry), then the entries' terms are added IN LIST ORDER into the candidates' potentials:
// lane c keeps candidate c's sum and step i adds entry i's term (broadcast with readlane) -- the
// same sequence of float64 additions as potential() (inference.py:55-71).
// ---------------------------------------------------------------------------------------------
struct HubEntry { uint32_t wid, d1; int cstar, A, B; double w; };

// lane's entry of round r under the value array `val`: loads, gathers, closes
template <typename VT>
__device__ __forceinline__ void hub_entry(const DevGraph<VT> &g, const uint8_t *lut, const uint32_t *base,
                                          int rows, int r, int M, const VT *val, bool want_w, HubEntry &en) {
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t *row = base + (size_t)(r * rows) * 64 + lane;
    en.wid = row[0];
    en.d1 = row[64];
    uint32_t mw[6];
    int x[6];
#pragma unroll
    for (int m = 0; m < 6; m++) {                       // M is wave-uniform
        mw[m] = m < M ? row[(2 + m) * 64] : NSK_GEN_NULL;
        const uint32_t id = mw[m] & NSK_GEN_NULL;
        x[m] = (m < M && id != NSK_GEN_NULL) ? (int)val[id] : 0;
    }
    en.w = want_w ? g.w[en.wid] : 0.0;
    const bool cat = (en.d1 & 15u) >= 6u;
    GenChain a;
    a.open();
#pragma unroll
    for (int m = 0; m < 6; m++)
        if (m < M) a.member(m == 0, cat, mw[m], x[m]);
    a.close(en.d1, lut, en.cstar, en.A, en.B);
}

// candidate `lane`'s potential of the hub over `val` (every lane returns its own candidate's sum)
template <typename VT>
__device__ __forceinline__ double hub_potentials(const DevGraph<VT> &g, const uint8_t *lut, const uint4 hd,
                                                 const VT *val) {
    const int lane = (int)(threadIdx.x & 63);
    const int n = (int)hd.y, M = (int)(hd.z & 0xFFu), rows = 2 + M;
    const uint32_t *base = g.hub_adj + hd.x;
    double pc = 0.0;
    for (int r = 0; r * 64 < n; r++) {
        HubEntry en;
        hub_entry(g, lut, base, rows, r, M, val, true, en);
        const double tA = en.w * (double)en.A, tB = en.w * (double)en.B;
        const int ks = (int)((en.d1 >> 14) & 15u);
        const int nlive = min(64, n - r * 64);
        for (int i = 0; i < nlive; i++) {               // list order
            const int s_cstar = __builtin_amdgcn_readlane(en.cstar, i);
            const int s_ks = __builtin_amdgcn_readlane(ks, i);
            const double s_tA = lane_value(tA, i), s_tB = lane_value(tB, i);
            const bool on = s_ks == 15 || s_ks == lane;
            const double t = on ? (lane == s_cstar ? s_tA : s_tB) : 0.0;       // +0.0 leaves the sum unchanged
            pc = pc + t;
        }
    }
    return pc;
}

// draw_sample (inference.py:36-52) from per-lane candidate potentials; every lane returns the value
__device__ __forceinline__ int hub_draw(double pc, int card, double u) {
    const int lane = (int)(threadIdx.x & 63);
    const double ek = nsk_exp(pc);
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

template <typename VT>
__device__ __forceinline__ void heavy_update_ep(const DevGraph<VT> &g, const uint8_t *lut, int p, const uint4 hd,
                                                int sample_evidence, int burnin, uint32_t k0, uint32_t k1,
                                                uint32_t s0, uint32_t s1) {
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info);
    const int v = g.p_vid[p];
    if (v < 0 || !(ev == 0 || sample_evidence)) return;
    const int card = NSK_INFO_CARD(info);
    const double pc = hub_potentials(g, lut, hd, g.val);
    const uint2 r = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const int nv = hub_draw(pc, card, u53(r.x, r.y));
    if ((threadIdx.x & 63) == 0) {
        g.val[p] = (VT)nv;
        if (!burnin) {
            const int base = g.p_cnt[p];
            if (card == 2) g.cnt[base] += nv;
            else g.cnt[base + nv] += 1;
        }
    }
}


template <typename VT, int MAXC>
__device__ __forceinline__ void gibbs_tile_general(const DevGraph<VT> &g, const uint8_t *lut, const uint4 *sp,
                                                   uint32_t tdw, uint32_t prog, uint32_t wrow, int p, bool valid,
                                                   int sample_evidence,
                                                   int burnin, uint32_t k0, uint32_t k1, uint32_t s0,
                                                   uint32_t s1) {
    const int len = (int)(tdw & 0xFFu), maxcard = (int)((tdw >> 12) & 15u);
    const uint32_t info = valid ? g.p_info[p] : (2u << 9);
    GenPot<MAXC> pot;
    pot.clear();
    auto on_entry = [&](uint32_t, double w, uint32_t d1, const GenChain &a, const GenChain &) {
        int cstar, A, B;
        a.close(d1, lut, cstar, A, B);
        pot.add(maxcard, d1, w, cstar, A, B);
    };
    if ((tdw >> 19) & 1u)                      // materialised weight rows (large weight tables)
        general_walk<VT, false, 1, true>(g, g.val, g.val, sp, len, (int)((tdw >> 16) & 7u), prog,
                                   g.adj_wt + (size_t)wrow * 64 + (threadIdx.x & 63), on_entry);
    else
        general_walk<VT, false, 0, true>(g, g.val, g.val, sp, len, (int)((tdw >> 16) & 7u), prog, nullptr, on_entry);
    const int ev = NSK_INFO_EV(info);
    if (!valid || !(ev == 0 || sample_evidence)) return;
    const int card = NSK_INFO_CARD(info);
    const uint2 rr = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const int nv = pot.draw(maxcard, card, u53(rr.x, rr.y));
    g.val[p] = (VT)nv;
    if (!burnin) {
        if (card == 2) g.cnt_pos[p] = (uint8_t)(g.cnt_pos[p] + nv);
        else g.cnt[g.p_cnt[p] + nv] += 1;
    }
}

template <typename VT>
__device__ __forceinline__ void fast_tile_update(const DevGraph<VT> &g, int pbegin, int pend, int wb_base,
                                                 int wave, int sample_evidence, int burnin, uint32_t k0,
                                                 uint32_t k1, uint32_t s0, uint32_t s1);

template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_fast(DevGraph<VT> g, int pbegin, int pend,
                                                          int wb_base, int nblocks,
                                                          const uint32_t *tile_list, int nlist,
                                                          int sample_evidence, int burnin,
                                                          uint32_t k0, uint32_t k1, uint32_t s0,
                                                          uint32_t s1) {
    const int lb = xcd_logical_block((int)blockIdx.x, nblocks);
    if (lb < 0) return;
    int wave = __builtin_amdgcn_readfirstlane(lb * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (tile_list) {                                      // list mode: the tiles outside segments
        if (wave >= nlist) return;
        wave = (int)__builtin_amdgcn_readfirstlane(tile_list[wave]);
    }
    fast_tile_update(g, pbegin, pend, wb_base, wave, sample_evidence, burnin, k0, k1, s0, s1);
}

// One uniform / shape / per-lane-header tile (tile index `wave` of the colour), one wave.
template <typename VT>
__device__ __forceinline__ void fast_tile_update(const DevGraph<VT> &g, int pbegin, int pend, int wb_base,
                                                 int wave, int sample_evidence, int burnin, uint32_t k0,
                                                 uint32_t k1, uint32_t s0, uint32_t s1) {
    const int lane = (int)(threadIdx.x & 63);
    const int p = pbegin + wave * 64 + lane;
    if (pbegin + wave * 64 >= pend) return;               // whole wave beyond the range
    const int v0 = p < pend ? g.p_vid[p] : -1;            // -1 also marks padding positions
    const bool valid = v0 >= 0;
    const uint32_t info = valid ? g.p_info[p] : 0u;
    const NSK_SCALAR uint32_t *tdp = (const NSK_SCALAR uint32_t *)(g.tiles + (wb_base + wave));
    const struct { uint32_t x, y, z, w; } td = {tdp[0], tdp[1], tdp[2], tdp[3]};
    const uint4 *sp = g.adj + td.x + lane;
    const int len = (int)td.y;
    // the tally byte is fetched now so that its latency overlaps the tile walk
    const uint8_t tally = (valid && !burnin) ? g.cnt_pos[p] : (uint8_t)0;

    double p0 = 0.0, p1 = 0.0;
    if (td.z == NSK_PAD_WORD) tile_potentials_dynamic(g, g.val, sp, len, p0, p1);
    else {
        const uint32_t kind = (td.w >> 8) & 7u;              // wave-uniform
        if (kind == 7u)
            tile_potentials_shape(g, g.val, sp, (int)(td.w & 0xFFu), td.z,
                                  *(const NSK_SCALAR uint32_t *)(g.tile_wrow + (wb_base + wave)), p0, p1);
        else if (kind == 4u) tile_potentials_uniform<VT, 4>(g, g.val, sp, len, td.z, p0, p1);
        else if (kind == 0u) tile_potentials_uniform<VT, 0>(g, g.val, sp, len, td.z, p0, p1);
        else if (kind == 2u) tile_potentials_uniform<VT, 2>(g, g.val, sp, len, td.z, p0, p1);
        else tile_potentials_uniform<VT, 3>(g, g.val, sp, len, td.z, p0, p1);
    }
    if (!valid) return;
    const int ev = NSK_INFO_EV(info);
    if (!(ev == 0 || sample_evidence)) return;
    const uint2 rr = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const double z0 = nsk_exp(p0);
    const double z1 = z0 + nsk_exp(p1);
    const double z = u53(rr.x, rr.y) * z1;
    const int nv = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    g.val[p] = (VT)nv;
    if (!burnin) g.cnt_pos[p] = (uint8_t)(tally + nv);
}

// The general tiles of one colour class: tiles [tile0, tile0 + ntiles) of the colour, one wave each.
// Blocks [0, hblocks) of the grid are hub blocks (one wave per position of [hb, he)), the next ones
// walk the general tiles, the last ones the colour's other tiles outside segments (rest_list): one
// launch per colour class instead of three.
template <typename VT, int MAXC>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_general(DevGraph<VT> g, int pbegin, int pend,
                                                             int wb_base, int tile0, int ntiles,
                                                             int nblocks, int hb, int he, int hblocks, int hub0,
                                                             const uint32_t *rest_list, int nrest,
                                                             int sample_evidence, int burnin,
                                                             uint32_t k0, uint32_t k1, uint32_t s0,
                                                             uint32_t s1) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[2048];
    load_gen_lut(lut);
    if ((int)blockIdx.x < hblocks) {                      // block-uniform
        const int hp = hb + (int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
        if (hp < he) {
            const NSK_SCALAR uint32_t *hdp = (const NSK_SCALAR uint32_t *)(g.hub_desc + hub0 + (hp - hb));
            const uint4 hd = {hdp[0], hdp[1], hdp[2], hdp[3]};
            if (hd.y) heavy_update_ep(g, lut, hp, hd, sample_evidence, burnin, k0, k1, s0, s1);
            else heavy_update(g, hp, sample_evidence, burnin, k0, k1, s0, s1);
        }
        return;
    }
    const int tblocks = 8 * ((nblocks + 7) / 8);
    if ((int)blockIdx.x >= hblocks + tblocks) {           // the colour's uniform / shape tiles outside segments
        const int i = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - hblocks - tblocks) * (NSK_BLOCK / 64) +
                                                     (int)(threadIdx.x >> 6));
        if (i < nrest)
            fast_tile_update(g, pbegin, pend, wb_base, (int)__builtin_amdgcn_readfirstlane(rest_list[i]),
                             sample_evidence, burnin, k0, k1, s0, s1);
        return;
    }
    const int lb = xcd_logical_block((int)blockIdx.x - hblocks, nblocks);
    if (lb < 0) return;
    const int lane = (int)(threadIdx.x & 63);
    const int t = __builtin_amdgcn_readfirstlane(lb * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (t >= ntiles) return;
    const int tile = tile0 + t;
    const int p = pbegin + tile * 64 + lane;
    const bool valid = p < pend && g.p_vid[p] >= 0;
    const NSK_SCALAR uint32_t *tdp = (const NSK_SCALAR uint32_t *)(g.tiles + (wb_base + tile));
    const uint32_t tdx = tdp[0], tdz = tdp[2], tdw = tdp[3];
    const uint32_t wrow = *(const NSK_SCALAR uint32_t *)(g.tile_wrow + (wb_base + tile));
    gibbs_tile_general<VT, MAXC>(g, lut, g.adj + tdx + lane, tdw, tdz, wrow, p, valid, sample_evidence, burnin,
                                 k0, k1, s0, s1);
}

// ---------------------------------------------------------------------------------------------
// Entry-parallel groups (nsk_compile.h ep_desc; DESIGN.md "entry-parallel groups").  A general tile
// walked by one lane per VARIABLE is a chain of dependent memory round trips -- stream words, member
// gathers, next entries -- as long as its longest lane and padded to its widest lane.  Here a
// workgroup owns 256 positions and alternates two phases:
//   1. one lane per LIST ENTRY: the entries of the 256 variables, sorted by member count, are rows of
//      64; a lane loads its entry's words (coalesced), gathers the weight and the <= 3 member values,
//      closes the chain facts exactly like a general tile (GenChain, gen_lut_entry) and leaves
//      (weight, owner candidate, cstar, A, B) in the LDS slot [position in the list][variable].
//      No lane waits for a previous entry; the next row's words are requested before this row's gathers.
//   2. one lane per VARIABLE: adds its entries' terms IN LIST ORDER from LDS -- the same float64
//      additions as potential() (inference.py:55-71) -- then draws and stores.
// LDS holds 8 list positions per variable (22 KB per workgroup: 7 workgroups per CU); a group whose
// variables have up to 16 entries takes two passes (its entries 8..15 are rows of their own).  The
// grid is resident: a workgroup walks the groups of its XCD's eighth of the colour.
// ---------------------------------------------------------------------------------------------
typedef unsigned int nsk_u32x2 __attribute__((ext_vector_type(2)));
#define NSK_EP_WID(w0) ((w0) & 0x7FFFFFFu)
#define NSK_EP_SLOT(w0, d1) ((((w0) >> 27) & 7u) * 256u + (((d1) >> 23) & 255u))
#define NSK_EP_LIST 8                      // list positions per variable held in LDS

struct EpRow { uint32_t w0, d1, m[3]; };

// sub-rows of the rows of one pass: classes M = 0..3 with the row counts of `rowsw` (8 bits each)
__device__ __forceinline__ int ep_pass_subrows(uint32_t rowsw) {
    return (int)(rowsw & 255u) * 2 + (int)((rowsw >> 8) & 255u) * 3 + (int)((rowsw >> 16) & 255u) * 4 + (int)(rowsw >> 24) * 5;
}
__device__ __forceinline__ int ep_pass_rows(uint32_t rowsw) {
    return (int)(rowsw & 255u) + (int)((rowsw >> 8) & 255u) + (int)((rowsw >> 16) & 255u) + (int)(rowsw >> 24);
}

// words of row r (wave-uniform) of a pass; M = its member count.  The number of loads does not depend on the row:
// a row with fewer than three member sub-rows reads its last sub-row again (the line is on its way already) and
// the words are replaced by the empty slot -- with a load count that varies from row to row the compiler cannot
// count the loads in flight and waits for ALL of them (s_waitcnt vmcnt(0)) at the first use of a row, which is
// where the prefetch of the next row had just been issued.
template <bool NT, bool MEMBERS, bool WT>
__device__ __forceinline__ void ep_load_row(const uint32_t *adj, uint32_t sub0, uint32_t rowsw, int r, EpRow &q, int &M,
                                            const double *wt, double &w) {
    const int n0 = (int)(rowsw & 255u), n1 = (int)((rowsw >> 8) & 255u), n2 = (int)((rowsw >> 16) & 255u);
    const int c1 = n0 + n1, c2 = c1 + n2;
    int sub;
    if (r < n0) { M = 0; sub = r * 2; }
    else if (r < c1) { M = 1; sub = n0 * 2 + (r - n0) * 3; }
    else if (r < c2) { M = 2; sub = n0 * 2 + n1 * 3 + (r - c1) * 4; }
    else { M = 3; sub = n0 * 2 + n1 * 3 + n2 * 4 + (r - c2) * 5; }
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t *row = adj + (size_t)(sub0 + (uint32_t)sub) * 64;
    const nsk_u32x2 h = NT ? __builtin_nontemporal_load((const nsk_u32x2 *)row + lane) : *((const nsk_u32x2 *)row + lane);
    q.w0 = h.x; q.d1 = h.y;
    if (WT) w = __builtin_nontemporal_load(wt + (size_t)r * 64 + lane);     // the entry's materialised weight
    q.m[0] = NSK_GEN_NULL; q.m[1] = NSK_GEN_NULL; q.m[2] = NSK_GEN_NULL;
    if (MEMBERS) {
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const int mm = M > m ? 2 + m : 1 + M;                          // (wave-uniform: the sub-row to read)
            const uint32_t x = NT ? __builtin_nontemporal_load(row + mm * 64 + lane) : row[mm * 64 + lane];
            q.m[m] = M > m ? x : NSK_GEN_NULL;
        }
    }
}

// The rows of one pass that belong to this wave (dealt round-robin over the four waves):
// fn(weight word, descriptor, chain over va, chain over vb, weight).  MEMBERS = false: only the
// (weight word, descriptor) pairs are read (the gradient pass of learning).
// WMODE: the entry's weight -- 0 not needed, 1 gathered from g.w (learning: weights change with every
// class), 2 read from the materialised rows `wt` (first row of this pass; inference).
// A wave takes U of its rows per step: their words were requested during the previous step,
// their gathers are issued together and waited for once.
// (measured on the 5M LR graph, per class: inference U = 1 58.5 us, 2 63.8, 3 64.9 -- the single-chain walk
// keeps 6 waves per SIMD; learning, whose rows carry two chains, U = 1 159.8, 2 149.4, 3 164.2)
// (round 5, once the loads of a row could be counted -- ep_load_row --: inference at 50M 468.5 -> 412.7 us per class;
// on top of that, rows two deep in flight -- words of row k + 2 and gathers of row k + 1 behind the arithmetic of
// row k, three register sets rotating through a loop unrolled by three, every wait a counted one -- 421.5 us, the
// 96-register variant of the small graphs spilling (5M: 49.6 -> 55.7 us); two rows per step 411.0 / 66.5 us:
// a step no longer waits for its own prefetch; what a group takes now is the serial structure of its pass.
// tools/sessions/r5_s20.sh, r5_s21.sh)
#define NSK_EP_WIN_ON false         // (value windows: measured, not in the default build -- nsk_compile.h ep_win)
// wa / wb: the group's value windows in LDS (ep_stage_window; `win`: in use) -- a member word whose id field is
// NSK_EP_WIN_BASE + o reads byte o of the window instead of gathering from the value array
template <typename VT, bool TWO, int WMODE, bool NT, bool MEMBERS, int U, typename FN>
__device__ __forceinline__ void ep_pass(const DevGraph<VT> &g, const VT *va, const VT *vb, uint32_t sub0,
                                        uint32_t rowsw, const double *wt, const signed char *wa, const signed char *wb, bool win, FN &&fn) {
    // (the wave's index as a SCALAR: row numbers, member counts, sub-row offsets and the branches on them are then
    // scalar work -- from threadIdx alone the compiler must assume they differ from lane to lane)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int total = ep_pass_rows(rowsw);
    if (wave >= total) return;
    EpRow cur[U], nxt[U];
    int Mc[U], Mn[U];
    double wc[U], wn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        Mc[u] = 0; Mn[u] = 0; wc[u] = 0.0; wn[u] = 0.0;
        if (wave + 4 * u < total) ep_load_row<NT, MEMBERS, WMODE == 2>(g.ep_adj, sub0, rowsw, wave + 4 * u, cur[u], Mc[u], wt, wc[u]);
    }
    for (int r = wave; r < total; r += 4 * U) {
        // this step's gathers FIRST, the next rows' words behind them (always the same number of loads: past the
        // last row the last row again): the gathers are then complete when all but those loads are (vmcnt = their
        // number), and the next rows' words have this step's arithmetic and the next step's gathers to arrive in
        double w[U];
        int xa[U][3], xb[U][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            w[u] = 0.0;
#pragma unroll
            for (int m = 0; m < 3; m++) { xa[u][m] = 0; xb[u][m] = 0; }
            if (r + 4 * u >= total) continue;                       // wave-uniform
            // (learning gathers the weight -- 68 % of the bytes a 50M-variable class fetches: the 8 MB table does
            // not stay in a 4 MB L2 next to the rows and values.  A non-temporal gather, so that the weight lines
            // would at least not evict the others, was measured (-DNSK_EP_W_NT, tools/sessions/r5_s02.sh): 5M LR
            // 120.6 -> 144.1 us per class, 50M 1088 -> 1306 -- the hot low ids DO hit the L2 when they may stay)
            w[u] = WMODE == 1 ? g.w[NSK_EP_WID(cur[u].w0)] : (WMODE == 2 ? wc[u] : 0.0);
            if (MEMBERS) {
#pragma unroll
                for (int m = 0; m < 3; m++)
                    if (Mc[u] > m) {                                // wave-uniform
                        const uint32_t id = cur[u].m[m] & NSK_GEN_NULL;
                        const uint32_t at = id == NSK_GEN_NULL ? 0u : id;
                        if (NSK_EP_WIN_ON && sizeof(VT) == 1 && win && id >= NSK_EP_WIN_BASE && id != NSK_GEN_NULL) {     // in the group's window
                            xa[u][m] = (int)wa[id - NSK_EP_WIN_BASE];
                            if (TWO) xb[u][m] = (int)wb[id - NSK_EP_WIN_BASE];
                        } else {
                            xa[u][m] = (int)va[at];
                            if (TWO) xb[u][m] = (int)vb[at];
                        }
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            ep_load_row<NT, MEMBERS, WMODE == 2>(g.ep_adj, sub0, rowsw, min(r + 4 * (U + u), total - 1), nxt[u], Mn[u], wt, wn[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (r + 4 * u >= total) continue;
            const bool cat = (cur[u].d1 & 15u) >= 6u;
            GenChain a, b;
            a.open();
            if (TWO) b.open();
            if (MEMBERS) {
#pragma unroll
                for (int m = 0; m < 3; m++)
                    if (Mc[u] > m) {
                        a.member(m == 0, cat, cur[u].m[m], xa[u][m]);
                        if (TWO) b.member(m == 0, cat, cur[u].m[m], xb[u][m]);
                    }
            }
            fn(cur[u].w0, cur[u].d1, a, b, w[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) { cur[u] = nxt[u]; Mc[u] = Mn[u]; wc[u] = wn[u]; }
    }
}

// The value window of group `gidx` (nsk_compile.h ep_win) into LDS: 16-byte chunks of the value array(s), one
// per thread and trip.  The caller's next barrier publishes it; nobody may still be reading the previous one.
template <typename VT, bool TWO>
__device__ __forceinline__ void ep_stage_window(const DevGraph<VT> &g, int gidx, nsk_u32x4 *wa, nsk_u32x4 *wb) {
    if (!NSK_EP_WIN_ON || sizeof(VT) != 1 || !g.ep_win) return;
    const NSK_SCALAR uint32_t *op = (const NSK_SCALAR uint32_t *)(g.ep_win_off + gidx);
    const uint32_t o0 = op[0], n = op[1] - o0;
    for (uint32_t i = threadIdx.x; i < n; i += NSK_BLOCK) {
        const uint32_t c = __builtin_nontemporal_load(g.ep_win + o0 + i);
        wa[i] = *((const nsk_u32x4 *)g.val + c);
        if (TWO) wb[i] = *((const nsk_u32x4 *)g.val_evid + c);
    }
}
#define NSK_EP_WIN_LDS(VT) 1          // (value windows are not in the default build: nsk_compile.h ep_win)

// ep_wt row <- the weights its entries name (run whenever weights may have changed, before an
// inference call): one workgroup per group, its rows dealt to the waves
static __global__ __launch_bounds__(NSK_BLOCK) void k_refresh_ep_weights(const uint4 *ep_desc, const uint32_t *ep_adj,
                                                                  const uint32_t *ep_wrow, const double *w,
                                                                  double *ep_wt) {
    const uint4 gd = ep_desc[blockIdx.x];
    const int lane = (int)(threadIdx.x & 63);
    uint32_t sub = gd.x;
    int row = (int)ep_wrow[blockIdx.x];
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t rowsw = pass ? gd.w : gd.y;
        const int total = ep_pass_rows(rowsw);
        for (int r = (int)(threadIdx.x >> 6); r < total; r += NSK_BLOCK / 64) {
            EpRow q;
            int M;
            double unused = 0.0;
            ep_load_row<false, false, false>(ep_adj, sub, rowsw, r, q, M, nullptr, unused);
            ep_wt[(size_t)(row + r) * 64 + lane] = w[NSK_EP_WID(q.w0)];
        }
        sub += (uint32_t)ep_pass_subrows(rowsw);
        row += total;
    }
}

// (cstar, A, B) of a closed entry in 8 bits; a cstar no candidate of a general variable (< 8) can
// equal is stored as 15
__device__ __forceinline__ uint32_t ep_facts(int cstar, int A, int B) {
    return ((uint32_t)cstar > 15u ? 15u : (uint32_t)cstar) | ((uint32_t)(A + 1) << 4) | ((uint32_t)(B + 1) << 6);
}

// The groups of a resident launch that this workgroup walks.  Groups are dealt to the XCDs (XCD =
// blockIdx & 7, how the hardware deals workgroups; for speed only) in chunks of 16 consecutive groups,
// round-robin: neighbouring groups read the same value lines (one L2), and every XCD gets its share
// of every region of the colour (the long-list groups come first).  `li` counts a workgroup's XCD-local
// groups; ep_group() maps it to the group or -1.
#ifndef NSK_EP_U_INF
#define NSK_EP_U_INF 1            // rows a wave takes per step in the inference launch (ep_pass)
#endif
#define NSK_EP_CHUNK 16
struct EpWalk { int li, lend, step, xcd; };
__device__ __forceinline__ EpWalk ep_walk(int ngroups, int hblocks, int gblocks) {
    const int xcd = (int)(blockIdx.x & 7);
    const int firstb = hblocks + ((xcd - (hblocks & 7) + 8) & 7);        // first group block on this XCD
    const int nbx = firstb < hblocks + gblocks ? (hblocks + gblocks - 1 - firstb) / 8 + 1 : 0;
    EpWalk wk;
    wk.xcd = xcd;
    wk.li = ((int)blockIdx.x - firstb) >> 3;
    wk.lend = ((ngroups + 8 * NSK_EP_CHUNK - 1) / (8 * NSK_EP_CHUNK)) * NSK_EP_CHUNK;
    wk.step = nbx;
    return wk;
}
__device__ __forceinline__ int ep_group(const EpWalk &wk, int li, int ngroups) {
    const int gi = ((li / NSK_EP_CHUNK) * 8 + wk.xcd) * NSK_EP_CHUNK + (li % NSK_EP_CHUNK);
    return gi < ngroups ? gi : -1;
}

template <typename VT>
__device__ __forceinline__ double block_hub_potentials(const DevGraph<VT> &g, const uint8_t *lut, const uint4 hd,
                                                       const VT *val, double *ws, uint16_t *fs) {
    // LDS per round of NSK_HUB_CHUNK entries: the two possible terms of an entry (weight x A, weight x B:
    // the products potential() forms) in ws[0 .. CHUNK) / ws[CHUNK .. 2 CHUNK), owner | cstar << 4 in fs
    constexpr int CHUNK = NSK_EP_LIST * 128;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int n = (int)hd.y, M = (int)(hd.z & 0xFFu), rows = 2 + M;
    const uint32_t *base = g.hub_adj + hd.x;
    double pc = 0.0;
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        const int cn = min(CHUNK, n - c0);
        __syncthreads();                                   // (wave 0 is done with the previous round)
        for (int r = wave; r * 64 < cn; r += NSK_BLOCK / 64) {
            HubEntry en;
            hub_entry(g, lut, base, rows, c0 / 64 + r, M, val, true, en);
            ws[r * 64 + lane] = en.w * (double)en.A;
            ws[CHUNK + r * 64 + lane] = en.w * (double)en.B;
            fs[r * 64 + lane] = (uint16_t)(((en.d1 >> 14) & 15u) | (((uint32_t)en.cstar > 15u ? 15u : (uint32_t)en.cstar) << 4));
        }
        __syncthreads();
        if (wave == 0)
            for (int i0 = 0; i0 < cn; i0 += 16) {          // list order; 16 entries' LDS reads in flight at a time
                double tA[16], tB[16];
                uint32_t f[16];
#pragma unroll
                for (int j = 0; j < 16; j++) {             // (entries past cn: the round's padding, owner 14, or
                    const int i = min(i0 + j, cn - 1);     //  a repeat that is masked below)
                    tA[j] = ws[i]; tB[j] = ws[CHUNK + i]; f[j] = fs[i];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int ks = (int)(f[j] & 15u), cstar = (int)((f[j] >> 4) & 15u);
                    const bool on = i0 + j < cn && (ks == 15 || ks == lane);
                    const double t = on ? (lane == cstar ? tA[j] : tB[j]) : 0.0;     // +0.0 leaves the sum unchanged
                    pc = pc + t;
                }
            }
    }
    return pc;                                             // meaningful in wave 0
}

template <typename VT>
__device__ __forceinline__ void block_hub_update(const DevGraph<VT> &g, const uint8_t *lut, int p, const uint4 hd,
                                                 double *ws, uint16_t *fs, int sample_evidence, int burnin,
                                                 uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1) {
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info);
    if (g.p_vid[p] < 0 || !(ev == 0 || sample_evidence)) return;             // (block-uniform)
    const int card = NSK_INFO_CARD(info);
    const double pc = block_hub_potentials(g, lut, hd, g.val, ws, fs);
    if ((threadIdx.x >> 6) != 0) return;
    const uint2 r = inf_words(k0, k1, (uint32_t)p, s0, s1);
    const int nv = hub_draw(pc, card, u53(r.x, r.y));
    if ((threadIdx.x & 63) == 0) {
        g.val[p] = (VT)nv;
        if (!burnin) {
            const int base = g.p_cnt[p];
            if (card == 2) g.cnt[base] += nv;
            else g.cnt[base + nv] += 1;
        }
    }
}

template <typename VT, int MAXC>
__device__ __forceinline__ void gibbs_ep_body(const DevGraph<VT> &g, int pbegin, int pend, int wb_base,
                                              int tile0, int ntiles, int ngroups, int group0, int gblocks,
                                              int hb, int he, int hblocks, int hub0, int nbh, int bh0,
                                              const uint32_t *rest_list, int nrest,
                                              int sample_evidence, int burnin,
                                              uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1) {
    __shared__ __attribute__((aligned(16))) double ws[NSK_EP_LIST * 256];
    __shared__ __attribute__((aligned(16))) uint16_t fs[NSK_EP_LIST * 256];
    __shared__ __attribute__((aligned(16))) uint8_t lut[2048];
    __shared__ nsk_u32x4 wina[NSK_EP_WIN_LDS(VT)];                 // the group's value window (ep_stage_window)
    load_gen_lut(lut);
    // blocks [0, nbh): one long-list hub each; [nbh, hblocks): one wave per hub position; then the
    // resident group blocks; then the colour's rest tiles
    if ((int)blockIdx.x < nbh) {
        const int hp = (int)__builtin_amdgcn_readfirstlane(g.bighub_pos[bh0 + (int)blockIdx.x]);
        const NSK_SCALAR uint32_t *hdp = (const NSK_SCALAR uint32_t *)(g.hub_desc + hub0 + (hp - hb));
        const uint4 hd = {hdp[0], hdp[1], hdp[2], hdp[3]};
        block_hub_update(g, lut, hp, hd, ws, fs, sample_evidence, burnin, k0, k1, s0, s1);
        return;
    }
    if ((int)blockIdx.x < hblocks) {                      // hub blocks: as in k_gibbs_general
        const int hp = hb + (int)((blockIdx.x - nbh) * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
        if (hp < he && !*(const NSK_SCALAR uint32_t *)((const uint32_t *)(g.hub_desc + hub0 + (hp - hb)) + 3)) {
            const NSK_SCALAR uint32_t *hdp = (const NSK_SCALAR uint32_t *)(g.hub_desc + hub0 + (hp - hb));
            const uint4 hd = {hdp[0], hdp[1], hdp[2], hdp[3]};
            if (hd.y) heavy_update_ep(g, lut, hp, hd, sample_evidence, burnin, k0, k1, s0, s1);
            else heavy_update(g, hp, sample_evidence, burnin, k0, k1, s0, s1);
        }
        return;
    }
    if ((int)blockIdx.x >= hblocks + gblocks) {           // the colour's uniform / shape tiles outside segments
        const int i = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - hblocks - gblocks) * (NSK_BLOCK / 64) +
                                                     (int)(threadIdx.x >> 6));
        if (i < nrest)
            fast_tile_update(g, pbegin, pend, wb_base, (int)__builtin_amdgcn_readfirstlane(rest_list[i]),
                             sample_evidence, burnin, k0, k1, s0, s1);
        return;
    }
    const EpWalk wk = ep_walk(ngroups, hblocks, gblocks);
    for (int li = wk.li; li < wk.lend; li += wk.step) {
        const int gi = ep_group(wk, li, ngroups);
        if (gi < 0) continue;
        const NSK_SCALAR uint32_t *gdp = (const NSK_SCALAR uint32_t *)(g.ep_desc + group0 + gi);
        const uint32_t gsub = gdp[0], grows0 = gdp[1], gmax = gdp[2], grows1 = gdp[3];
        const int ne = (int)(gmax & 255u);
        // this lane's variable (phase 2): requested now, needed after the entries
        const int tile = tile0 + 4 * gi + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (a scalar: see ep_pass)
        const bool tile_ok = tile < tile0 + ntiles;                       // wave-uniform
        const int p = pbegin + tile * 64 + (int)(threadIdx.x & 63);
        const bool valid = tile_ok && p < pend && g.p_vid[p] >= 0;
        const uint32_t info = valid ? g.p_info[p] : (2u << 9);
        const uint32_t tdw = tile_ok ? *(const NSK_SCALAR uint32_t *)((const uint32_t *)(g.tiles + (wb_base + tile)) + 3) : 0u;
        const int maxcard = (int)((tdw >> 12) & 15u);
        GenPot<MAXC> pot;
        pot.clear();
        uint32_t sub = gsub, wrow = *(const NSK_SCALAR uint32_t *)(g.ep_wrow + group0 + gi);
        for (int pass = 0; pass * NSK_EP_LIST < ne; pass++) {
            const uint32_t rowsw = pass ? grows1 : grows0;
            __syncthreads();                               // (the previous sums have been read)
            if (pass == 0) ep_stage_window<VT, false>(g, group0 + gi, wina, wina);
            // (a group of two-candidate variables could keep, per slot, the weight and two selectors in {-1, 0, 1} -- what
            // each candidate's sum takes of it, decided by the entry's lane -- so that the variable's lane adds w, -w or
            // nothing: 16 instead of 45 vector instructions per list position and variable.  Measured: 50M LR 408.6
            // against 404.2 us per class, weighted boolean graph 31.7 / 31.3, 5M LR 49.9 / 50.1 -- the arithmetic of
            // phase 2 is not what a group waits for; tools/sessions/r5_s25.sh; not kept)
            for (int i = (int)threadIdx.x; i < NSK_EP_LIST * 128; i += NSK_BLOCK)      // slots no entry writes:
                ((uint32_t *)fs)[i] = 14u | (14u << 16);                                   // owned by no candidate
            __syncthreads();
            auto entry_done = [&](uint32_t w0, uint32_t d1, const GenChain &a, const GenChain &, double w) {
                    int cstar, A, B;
                    a.close(d1, lut, cstar, A, B);
                    const uint32_t ks = (d1 >> 14) & 15u;
                    if (ks != 14u) {
                        const uint32_t slot = NSK_EP_SLOT(w0, d1);
                        ws[slot] = w;
                        fs[slot] = (uint16_t)(ks | (ep_facts(cstar, A, B) << 4));
                    }
                };
            ep_pass<VT, false, 2, true, true, NSK_EP_U_INF>(g, g.val, g.val, sub, rowsw, g.ep_wt + (size_t)wrow * 64,
                (const signed char *)wina, (const signed char *)wina, g.ep_win != nullptr, entry_done);
            __syncthreads();
            const int nacc = min(NSK_EP_LIST, ne - pass * NSK_EP_LIST);
            if (tile_ok)
                for (int o = 0; o < nacc; o++) {
                    const uint32_t f = fs[o * 256 + (int)threadIdx.x];
                    const double w = ws[o * 256 + (int)threadIdx.x];
                    pot.add_ks(maxcard, (int)(f & 15u), w, (int)((f >> 4) & 15u), (int)((f >> 8) & 3u) - 1, (int)((f >> 10) & 3u) - 1);
                }
            sub += (uint32_t)ep_pass_subrows(rowsw);
            wrow += (uint32_t)ep_pass_rows(rowsw);
        }
        const int ev = NSK_INFO_EV(info);
        if (valid && (ev == 0 || sample_evidence)) {
            const int card = NSK_INFO_CARD(info);
            const uint2 rr = inf_words(k0, k1, (uint32_t)p, s0, s1);
            const int nv = pot.draw(maxcard, card, u53(rr.x, rr.y));
            g.val[p] = (VT)nv;
            if (!burnin) {
                if (card == 2) g.cnt_pos[p] = (uint8_t)(g.cnt_pos[p] + nv);
                else g.cnt[g.p_cnt[p] + nv] += 1;
            }
        }
    }
}

#define NSK_EP_PARAMS DevGraph<VT> g, int pbegin, int pend, int wb_base, int tile0, int ntiles, int ngroups, int group0, int gblocks, \
                      int hb, int he, int hblocks, int hub0, int nbh, int bh0, const uint32_t *rest_list, int nrest, \
                      int sample_evidence, int burnin, uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1
#define NSK_EP_FORWARD g, pbegin, pend, wb_base, tile0, ntiles, ngroups, group0, gblocks, hb, he, hblocks, hub0, nbh, bh0, rest_list, nrest, \
                       sample_evidence, burnin, k0, k1, s0, s1
// (diagnostic builds: waves per SIMD of the inference launches.  Six -- 80 vector registers, 12-28 bytes of scratch --
// against the five the kernels reach on their own: 5M LR 49.1 -> 50.7 us per class, weighted boolean graph 30.7 -> 32.1,
// 50M LR 402 -> 444; tools/sessions/r5_s34.sh)
#ifdef NSK_EP_WPE_I
#define NSK_EP_ATTR_I __attribute__((amdgpu_waves_per_eu(NSK_EP_WPE_I, NSK_EP_WPE_I)))
#else
#define NSK_EP_ATTR_I
#endif
template <typename VT, int MAXC>
__global__ __launch_bounds__(NSK_BLOCK) NSK_EP_ATTR_I void k_gibbs_ep(NSK_EP_PARAMS) { gibbs_ep_body<VT, MAXC>(NSK_EP_FORWARD); }
// The same with the vector registers capped at 96 (5 waves per SIMD; the categorical kernel had 102 registers and 4
// waves until the wave index became a scalar -- round 5: it now has 90 and the cap is a formality): for
// graphs whose value array stays in the L2s the gathers are L2 hits and a fifth wave hides more of their
// latency (5M LR graph: 55.4 -> 51.2 us per class); beyond them it is slower (50M LR graph: 472 -> 485 us) --
// the launch picks by the size of the value array.
template <typename VT, int MAXC>
#ifndef NSK_EP_WPE_I
#define NSK_EP_WPE_I 5
#endif
__global__ __launch_bounds__(NSK_BLOCK) __attribute__((amdgpu_waves_per_eu(NSK_EP_WPE_I, NSK_EP_WPE_I))) void k_gibbs_ep_w5(NSK_EP_PARAMS) {
    gibbs_ep_body<VT, MAXC>(NSK_EP_FORWARD);
}
#undef NSK_EP_PARAMS
#undef NSK_EP_FORWARD

// Homogeneous segments: runs of consecutive uniform tiles with one program, slot count, kind and
// evidence flag (the shape-class layout of nsk_compile.cpp makes whole classes such runs).  Up to
// NSK_SEG_MAX segments of one (kind, chunk count) share a launch; everything the descriptor-driven
// kernel fetches per tile comes from the kernel-argument table here, and the body is straight
// line: ids, 16-byte member loads, byte gathers, compares, draw, store.
#define NSK_SEG_MAX 8
#define NSK_NO_STREAM 0xFFFFFFFFu          // SegEntry.aff_off of a segment without implicit adjacency
struct SegEntry {                         // 48 bytes: two scalar loads per tile (pair)
    int tile_start;                       // first tile of the segment in this launch's numbering
    int pos0;                             // position of the segment's first lane
    uint32_t adj_off;                     // stream offset (16-byte units) of its first tile
    uint32_t prog;                        // slot program
    uint32_t zoff;                        // draw-table launches: first entry of the program's table
    uint32_t zmask_ev;                    // (1 << member slots) - 1 | (uint8) common isEvidence << 8 | quad generator scheme << 16
    uint32_t ntiles_lead;                 // table launches: tiles of the segment | lead << 30 (dead virtual
                                          //   tiles in front: the segment does not start a quad)
    uint32_t aff_off;                     // first entry of the segment's tiles in seg_aff, NSK_NO_STREAM: none
    uint32_t push_off;                    // fused boundary exchange (TabP2P): first row of the segment's tiles in the
                                          //   push map, NSK_NO_STREAM: no tile of the segment touches the boundary
    uint32_t wide_off;                    // table launches: first dword of the quad descriptors (seg_wide) from the quad of
                                          //   pos0 on, NSK_NO_STREAM: none (int32 values, NSK_NO_WIDE)
    uint32_t pad_[2];
};
struct SegTable {
    int n, ntiles;                        // segments, tiles of the launch; e[i].tile_start = ntiles for i >= n
    int wide, pad_;                       // table launches: some segment has quad descriptors (SegEntry.wide_off)
    SegEntry e[NSK_SEG_MAX];
};

// segment of launch-tile T (wave-uniform): the first segment is the common case, the others are
// found by a short scalar scan
__device__ __forceinline__ int seg_of_tile(const SegTable &tab, int T) {
    int sidx = 0;
    if (T >= tab.e[1].tile_start)                           // (= ntiles when the launch has one segment)
        for (int i = 1; i < tab.n && T >= tab.e[i].tile_start; i++) sidx = i;
    return sidx;
}

// ---------------------------------------------------------------------------------------------
// Draw tables.  For a uniform program whose lanes read binary members only, the potentials of the
// two candidates -- hence z0 = exp(p0), z1 = z0 + exp(p1) -- depend on the <= 8 neighbour bits
// alone.  draw_sample's decision (inference.py:49-52) is  value = 0  iff  z0 >= fl(u * z1)  (or the
// second comparison fails, which needs a NaN);  u = k * 2^-53 is exact for the 53-bit integer k of
// the generator, and fl(u * z1) is non-decreasing in k, so the k that give 0 are a prefix [0, K]:
// K is found per neighbourhood by bisection over k WITH THE VERY SAME float64 OPERATIONS the
// sampling kernels and the oracle use (slot_step sums, nsk_exp, u53's product), and the sweep
// kernels compare integers.  Bit-identical to the exp-per-update path by construction (both are
// tested against the oracle).  sat0 / sat1: bit j = slot j's entry is satisfied for candidate 0 / 1
// (the learning kernels' gradient bits).
// ---------------------------------------------------------------------------------------------
struct ZProgDev { uint32_t prog, nslots, off, pad; };

__device__ __forceinline__ int draw_from_z(double z0, double z1, unsigned long long k) {
    const double u = u53((uint32_t)(k >> 26) << 5, (uint32_t)(k & 0x3FFFFFFull) << 6);
    const double z = u * z1;
    return (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
}

// One table entry: the program's slot algebra for neighbourhood `idx` with the per-slot terms
// delivered by term(j, thi, tlo); K by a galloping search around z0 / z1 * 2^53 (every probe is
// the real decision function, so the result is exact however poor the guess).
template <typename TERM>
__device__ __forceinline__ uint4 ztab_entry(const uint32_t *pp, uint32_t nslots, uint32_t idx, TERM &&term) {
    double p0 = 0.0, p1 = 0.0;
    uint32_t sat0 = 0, sat1 = 0;
    SlotState st = {0, true, false, true};
    for (uint32_t j = 0; j < nslots; j++) {
        const uint32_t s = pp[j];
        const int x = (int)((idx >> j) & 1u);
        const bool F = (s >> 27) & 1u, ig = (s >> 29) & 1u;          // slot_step / slot_sat, spelled out
        const uint32_t code = (s >> 24) & 7u;
        const bool nz = ig || (x != 0), one = !ig && (x == 1);
        st.alleq = F || (st.alleq && (x == st.first));
        st.allnz = (F || st.allnz) && nz;
        st.any1 = (!F && st.any1) || one;
        st.first = F ? x : st.first;
        const bool isEq = code == 4u, isAnd = code == 3u || code == 1u, isOr = code == 2u;
        const bool b0 = (isEq && st.alleq && (ig || st.first == 0)) || (isOr && st.any1);
        const bool b1 = (isEq && st.alleq && (ig || st.first == 1)) || (isAnd && st.allnz) || isOr;
        double thi, tlo;
        term(j, s, thi, tlo);
        p0 = p0 + (b0 ? thi : tlo);
        p1 = p1 + (b1 ? thi : tlo);
        sat0 |= (b0 ? 1u : 0u) << j;
        sat1 |= (b1 ? 1u : 0u) << j;
    }
    const double z0 = nsk_exp(p0);
    const double z1 = z0 + nsk_exp(p1);
    const unsigned long long top = (1ull << 53) - 1;
    // invariant: draw(lo) == 0 (draw(0) == 0 always: z = +0 or NaN), draw(hi) == 1 or hi == top + 1
    unsigned long long lo = 0, hi = top + 1;
    const double guess = (z0 / z1) * 9007199254740992.0;
    unsigned long long kg = (guess >= 0.0 && guess < 9007199254740992.0) ? (unsigned long long)guess : 0ull;
    if (kg > top) kg = top;
    if (draw_from_z(z0, z1, kg) == 0) {
        lo = kg;
        for (unsigned long long stp = 1; lo + stp <= top; stp <<= 1) {        // gallop up
            if (draw_from_z(z0, z1, lo + stp) == 0) lo += stp; else { hi = lo + stp; break; }
        }
        if (hi == top + 1) { if (draw_from_z(z0, z1, top) == 0) lo = top; else hi = top; }
    } else {
        hi = kg;
        for (unsigned long long stp = 1; ; stp <<= 1) {                        // gallop down
            const unsigned long long probe = hi > stp ? hi - stp : 0ull;
            if (draw_from_z(z0, z1, probe) == 0) { lo = probe; break; }
            hi = probe;
        }
    }
    while (lo < top && hi - lo > 1) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        if (draw_from_z(z0, z1, mid) == 0) lo = mid; else hi = mid;
    }
    // entry: the threshold's top 27 bits, its low 26 bits, the satisfied bits
    return uint4{(uint32_t)(lo >> 26), (uint32_t)(lo & 0x3FFFFFFull), sat0 | (sat1 << 8), 0u};
}

static __global__ __launch_bounds__(NSK_BLOCK) void k_refresh_ztab(const ZProgDev *zp, const uint32_t *tile_hdr,
                                                            const double *prog_w, uint4 *ztab) {
    const ZProgDev z = zp[blockIdx.x];
    const uint32_t idx = threadIdx.x;
    if (idx >= (1u << z.nslots)) return;
    const double *tw = prog_w + 2 * (size_t)z.prog;
    ztab[z.off + idx] = ztab_entry(tile_hdr + z.prog, z.nslots, idx,
                                   [&](uint32_t j, uint32_t, double &thi, double &tlo) { thi = tw[2 * j]; tlo = tw[2 * j + 1]; });
}

// the generator's 53-bit integer: u53(a, b) == k * 2^-53 exactly
__device__ __forceinline__ unsigned long long k53(uint32_t a, uint32_t b) {
    return ((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6);
}

// the 53-bit threshold of a table entry
__device__ __forceinline__ unsigned long long ztab_K(const uint4 &e) { return ((unsigned long long)e.x << 26) | (unsigned long long)e.y; }

// Homogeneous segments whose programs have draw tables: ids, byte gathers, bit pack, one 8-byte
// table read, integer compare.  No float64 arithmetic, no p_vid.
// * Resident grid (8 waves per SIMD): the waves loop over their work units, so the per-wave set-up --
//   kernel-argument loads, the Philox key schedule, base pointers -- is paid once per wave, and XCD x
//   walks the x-th eighth of the units (same locality as xcd_logical_block).
// * A QUAD is four tiles at positions 256 m .. 256 m + 255: lane l holds generator ids q, q + 64, q + 128,
//   q + 192, whose high words are the four words of ONE Philox block (nsk_device.h quad_block) -- one
//   Philox evaluation per lane decides four updates by comparing the top 27 bits with the tabulated
//   threshold's; the low words (a second block) are computed only when some lane ties (2^-27 per update).
//   The host numbers a launch's tiles "virtually": every segment starts on a quad boundary (lead = 1..3
//   dead tiles in front when its first tile is not the first of a quad).
//   A work unit is a quad (split = 0: big launches) or one of its two tile pairs (split = 1: launches
//   small enough that every pair gets a wave of its own; the block is then evaluated by both waves).
// * Padding lanes (class ends) sample like any other lane into their own, never-read, position; the
//   tally fold skips them.
// Round 2/3: instruction issue bounds the kernel, 63 of its 107 vector instructions
// per tile pair were the Philox rounds.
// Buffer addressing for the table kernels: base (128-bit descriptor in scalar registers) + a SCALAR offset + a
// per-lane offset.  A tile's member run "base of the slot + lane", its tally bytes and its stores are then
// addressed without a single vector instruction (the flat form cost one v_add per gather and, for the stores
// that a dead tile of a quad must not perform, a 64-bit add and two selects each: 64 of the 175 vector
// instructions of a quad, ISA count round 5), and a store with an out-of-range lane offset is dropped by the
// hardware's bounds check -- which is how dead tiles skip theirs.
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void *)nullptr, (short)0, 0, 0)) nsk_rsrc;
#define NSK_BUF_OOB 0xFFFFFFFFu            // >= num_records of every descriptor below
__device__ __forceinline__ nsk_rsrc nsk_make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, (short)0, (int)0xFFFFFFFFu, 0x00020000);
}
template <typename VT> __device__ __forceinline__ uint32_t nsk_buf_ld(nsk_rsrc r, uint32_t voff, uint32_t soff);      // low byte of element
template <> __device__ __forceinline__ uint32_t nsk_buf_ld<signed char>(nsk_rsrc r, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)voff, (int)soff, 0);
}
template <> __device__ __forceinline__ uint32_t nsk_buf_ld<int32_t>(nsk_rsrc r, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)(voff * 4u), (int)(soff * 4u), 0) & 0xFFu;
}
// system-coherent flavours (cache policy sc0 sc1): what a peer wrote is read from / what a peer will read is
// written through to memory, whatever this device's caches hold -- no acquire / release fence (a system-scope fence
// invalidates / writes back whole caches: with one per border wave a shard's class launch took 27 instead of 7 us)
#define NSK_AUX_SYS 17
template <typename VT> __device__ __forceinline__ uint32_t nsk_buf_ld_sys(nsk_rsrc r, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ uint32_t nsk_buf_ld_sys<signed char>(nsk_rsrc r, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)voff, (int)soff, NSK_AUX_SYS);
}
template <> __device__ __forceinline__ uint32_t nsk_buf_ld_sys<int32_t>(nsk_rsrc r, uint32_t voff, uint32_t soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)(voff * 4u), (int)(soff * 4u), NSK_AUX_SYS) & 0xFFu;
}
template <typename VT> __device__ __forceinline__ void nsk_buf_st_sys(nsk_rsrc r, uint32_t voff, int x);
template <> __device__ __forceinline__ void nsk_buf_st_sys<signed char>(nsk_rsrc r, uint32_t voff, int x) {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)x, r, (int)voff, 0, NSK_AUX_SYS);
}
template <> __device__ __forceinline__ void nsk_buf_st_sys<int32_t>(nsk_rsrc r, uint32_t voff, int x) {
    __builtin_amdgcn_raw_buffer_store_b32((unsigned int)x, r, (int)(voff == NSK_BUF_OOB ? voff : voff * 4u), 0, NSK_AUX_SYS);
}
template <typename VT> __device__ __forceinline__ void nsk_buf_st(nsk_rsrc r, uint32_t voff, uint32_t soff, int x);
template <> __device__ __forceinline__ void nsk_buf_st<signed char>(nsk_rsrc r, uint32_t voff, uint32_t soff, int x) {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)x, r, (int)voff, (int)soff, 0);
}
template <> __device__ __forceinline__ void nsk_buf_st<int32_t>(nsk_rsrc r, uint32_t voff, uint32_t soff, int x) {
    __builtin_amdgcn_raw_buffer_store_b32((unsigned int)x, r, (int)(voff == NSK_BUF_OOB ? voff : voff * 4u), (int)(soff * 4u), 0);
}

// ---- fused boundary exchange of a shard's table launches (N ranks, peer to peer; nsk_api.hip p2p_fuse_plan) ----
// The shard's ghosts are read straight from its receive block of the exchange allocation (parity of the last
// exchange) and the border tiles write their new values into the readers' blocks themselves (the other parity):
// a sweep of a shard is its class launches and nothing else -- no push, wait or unpack kernels.  Segments are
// split so that tiles that touch the boundary (own a value a peer reads, or read a ghost) form segments of their
// own (SegEntry.push_off): their waves wait for the peers' flags of the previous exchange before they read,
// interior tiles never wait; the wave that completes the sweep's last border tile raises this rank's flag at the
// peers.  Ghost values are one sweep old inside a sweep, as with the exchange kernels (and the reference's
// distributed loop, salt/src/numbskull_master.py:165-224): same samples, bit for bit.
struct TabP2P {
    const void *mine;                   // this rank's exchange allocation (flags | receive blocks | ...)
    void *peer[16];                     // the peers' allocations
    unsigned long long dtotal[16];      // length of peer q's per-chain receive block
    const uint32_t *push_map;           // per border tile and lane: reader << 28 | index in the reader's block; NSK_NO_STREAM
    unsigned int *counter;              // border tiles done in the running sweep
    unsigned int *err;
    const unsigned long long *tag_base; // captured launches: tag = tag_base[1] + tag
    unsigned long long timeout_ticks;
    uint32_t ghost_lo, nrecv;           // internal id of the first ghost = element 0 of the receive block
    uint32_t border_total;              // border tiles of one sweep, all classes
    uint32_t tag;                       // this sweep's exchange tag
    uint32_t peer_mask;
    int world, me;
    int wait;                           // this launch's border tiles wait for the peers' flags (the sweep's first class:
                                        //   later classes find them raised -- the stream is in order)
};
__host__ __device__ inline size_t nsk_p2p_recv_off_(int world) { return ((size_t)(4 * world) * 4 + 255) / 256 * 256; }   // = nsk_p2p_recv_off

#ifndef NSK_TAB_BATCH
#define NSK_TAB_BATCH 4       // tiles of a quad whose loads are in flight together (4: the whole quad; 2: pair by pair)
#endif
// NT consecutive tiles of one quad, first tile = segment tile `t0` (word `w0` of the quad's blocks): every
// load of the NT tiles is requested before the first draw, the draws' stores come last
// (PK: the launch keeps the tally INSIDE the value bytes -- k_gibbs_seg_tabw's packed mode, below: bit 0 the value, bits
//  1-7 the sweeps it was 1 since the last unpack -- so a member's byte is masked, the lane's own byte is its tally and the
//  store carries both)
template <typename VT, int NCH, int NT, bool PK = false>
__device__ __forceinline__ void tab_tiles(const DevGraph<VT> &g, const SegEntry &en, int t0, int w0, int lane, int burnin,
                                          const u32x4 &ra, u32x4 &rb, bool &have_b, uint32_t qb,
                                          uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1, bool ws) {
    const int nt = (int)(en.ntiles_lead & 0x3FFFFFFFu);
    const bool haff = en.aff_off != NSK_NO_STREAM;                      // implicit adjacency (nsk_compile.h seg_aff)
    const uint32_t zoff = en.zoff, zmask = en.zmask_ev & 0xFFu;
    bool live[NT];
    int p[NT], pl[NT], tt[NT];
    uint8_t tally[NT];
    uint32_t id[NT][4 * NCH], ab[NT][4 * NCH];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int t = t0 + k;
        live[k] = t >= 0 && t < nt;                                     // wave-uniform
        tt[k] = live[k] ? t : (t < 0 ? 0 : nt - 1);                     // (a dead tile reads a real one's data)
        p[k] = en.pos0 + t * 64 + lane;
        pl[k] = en.pos0 + tt[k] * 64 + lane;
        ab[k][0] = NSK_NO_STREAM;
    }
    if (haff) {                        // the tiles' slot bases: scalar loads, issued together
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const NSK_SCALAR uint32_t *ap = (const NSK_SCALAR uint32_t *)(g.seg_aff + en.aff_off + (size_t)tt[k] * NCH);
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) ab[k][j] = ap[j];
        }
    }
    // (the tally bytes in front of the gathers: the wait the stream words of a non-affine segment need -- vmcnt(0)
    // where the two paths meet -- then waits for them as well before an affine tile's gathers are issued.  Requesting
    // them BEHIND the gathers, here and for k_learn_seg_tab's evidence values, changed nothing: 10M grid 12.7 / 12.7 us
    // per class, learning 23.4 / 23.4, 40M grid 40.3 against 39.1 -- tools/sessions/r5_s23.sh; not kept)
#pragma unroll
    for (int k = 0; k < NT; k++) tally[k] = burnin ? (uint8_t)0 : (PK ? (uint8_t)g.val[pl[k]] : g.cnt_pos[pl[k]]);
#pragma unroll
    for (int k = 0; k < NT; k++) {
        if (ab[k][0] != NSK_NO_STREAM) {                                // wave-uniform: member = base + lane
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) id[k][j] = ab[k][j] + (uint32_t)lane;
        } else {
            const uint4 *sp = g.adj + en.adj_off + (size_t)tt[k] * (64 * NCH) + lane;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint4 w = sp[c * 64];
                id[k][4 * c] = w.x; id[k][4 * c + 1] = w.y; id[k][4 * c + 2] = w.z; id[k][4 * c + 3] = w.w;
            }
        }
    }
    // member values -> neighbourhood bits -> table entries (the table kernels run only while every value on
    // the device lies in its domain -- values_regular -- and their members are binary: a value IS its bit)
    uint32_t idx[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        idx[k] = 0;
#pragma unroll
        for (int j = 0; j < 4 * NCH; j++) idx[k] |= ((uint32_t)(uint8_t)g.val[id[k][j]] & (PK ? 1u : 0xFFu)) << j;
    }
    uint2 e[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) e[k] = *(const uint2 *)(g.ztab + zoff + (idx[k] & zmask));
    // draws: the high 27 bits decide unless they tie with the threshold's
    uint32_t hi[NT];
    int nv[NT];
    bool tie = false;
    // (ws, wave-uniform: the quad is a WIDE one -- nsk_compile.h seg_wide -- that this launch samples tile by tile, a
    // shard's run boundary cuts it: its positions keep the wide generator scheme, nsk_device.h wide_word_of_tile)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        hi[k] = (ws ? wide_word_of_tile(ra, (uint32_t)(w0 + k), (uint32_t)lane) : word_of(ra, (uint32_t)(w0 + k))) >> 5;
        nv[k] = hi[k] > e[k].x ? 1 : 0;
        tie = tie || hi[k] == e[k].x;
    }
    if (__builtin_expect(__any(tie), 0)) {
        if (!have_b) { rb = philox4x32(k0, k1, qb, 3u, s0, s1); have_b = true; }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const uint32_t lo = (ws ? wide_word_of_tile(rb, (uint32_t)(w0 + k), (uint32_t)lane) : word_of(rb, (uint32_t)(w0 + k))) >> 6;
            if (hi[k] == e[k].x) nv[k] = lo > e[k].y ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < NT; k++) {
        VT *dst = live[k] ? g.val + p[k] : (VT *)g.sink + lane;         // (no branch: see seg_of_tile's note)
        if (PK) {
            *dst = (VT)(uint8_t)((burnin ? 0u : (uint32_t)(tally[k] & 0xFEu)) + 3u * (uint32_t)nv[k]);   // tally + value, value
            continue;
        }
        *dst = (VT)nv[k];
        if (!burnin) {
            uint8_t *td = live[k] ? g.cnt_pos + p[k] : g.sink + 256 + lane;
            *td = (uint8_t)(tally[k] + nv[k]);
        }
    }
}

// ---- wide quads (nsk_compile.h seg_wide; int8 values): one lane samples FOUR consecutive positions ----
// Lane l of the wave takes positions p0 + 4 l .. p0 + 4 l + 3 of the quad at p0 (a multiple of 256).  Per member slot
// ONE dword load at (base of the slot) + 4 l brings the four neighbour bytes (256 contiguous bytes per wave-instruction:
// a quarter of the memory instructions of the tile-by-tile body, each four times as wide); the four neighbourhoods are
// packed by the same three shift-ors that used to pack one (SIMD within a register: a value is its bit); the four
// thresholds come from the block's copy of the draw table in LDS (their top 27 bits: what decides all but 2^-27 of the
// draws); the four new values and the four tally bytes leave as one dword each.  The lane's four draws are the four
// words of its own Philox block (the WIDE generator scheme, nsk_device.h wide_block).  A position whose member is not
// at base + offset -- the end cell of a grid row -- is patched from the quad's exception list by its lane.
typedef uint32_t nsk_u32_una __attribute__((aligned(1)));
#define NSK_ZT_BITS(NCH) (4 * (NCH))                  // log2 of the LDS table entries kept per segment of a launch
template <int NCH>
struct WideTrip { uint32_t x[4 * NCH], tally; };      // what a wide quad asks memory for: four neighbour bytes per slot, four tally bytes
// the requests of the quad at p0 (bases: its descriptor's, or any readable ones for a trip that will not be finished)
template <int NCH, int MODE>
__device__ __forceinline__ void wide_issue(const DevGraph<signed char> &g, int p0, const uint32_t (&base)[4 * NCH], int lane,
                                           WideTrip<NCH> &t) {
    // (32-bit offsets from the arrays' bases: `global_load_dword v, v_offset, s[base]` -- one vector add per load, no
    // 64-bit scalar pair per base)
    const uint32_t l4 = 4u * (uint32_t)lane;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) t.x[j] = *(const nsk_u32_una *)((const char *)g.val + (base[j] + l4));
    // the quad's tally bytes: its own value bytes in packed mode (MODE 2), else the position tally (MODE 0; 1 = burn-in)
    t.tally = 0;
    if (MODE == 2) t.tally = *(const uint32_t *)((const char *)g.val + ((uint32_t)p0 + l4));
    else if (MODE == 0) t.tally = *(const uint32_t *)((const char *)g.cnt_pos + ((uint32_t)p0 + l4));
}
// ... and the rest of the trip: draws, look-ups, stores.  wd: the quad's descriptor (nsk_compile.h seg_wide)
template <int NCH, int MODE>
__device__ __forceinline__ void wide_finish(const DevGraph<signed char> &g, uint32_t zoff, int p0,
                                            uint32_t exc0, uint32_t nexc, uint32_t smask, const WideTrip<NCH> &t, const uint32_t *zt,
                                            int lane, uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1,
                                            const char NSK_SCALAR *wexc_slot = nullptr, const PhiloxKeys *pk = nullptr) {
    // (the block is evaluated while the loads are in flight: without the fences the scheduler puts it behind the waits)
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t blk = (uint32_t)(p0 >> 2) + (uint32_t)lane;              // wide_block: ((p0 >> 8) << 6) | lane
    const u32x4 ra = pk ? philox4x32_keyed(*pk, blk, 2u, s0, s1) : philox4x32(k0, k1, blk, 2u, s0, s1);
    asm volatile("" :: "v"(ra.x), "v"(ra.y), "v"(ra.z), "v"(ra.w));       // (the words exist HERE: the optimiser sinks them to their uses otherwise)
    __builtin_amdgcn_sched_barrier(0);
    uint32_t idx4 = 0;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) idx4 |= (t.x[j] & 0x01010101u) << j;  // (a member's value is its bit; what a lane reads for a
    idx4 &= smask * 0x01010101u;                                            //  position that is an exception may be any value: masked)
    for (uint32_t e = 0; e < nexc; e++) {                                   // scalar loop, rare: the odd cells of the quad
        // (wexc_slot: where the kernel arguments hold the exception array's address -- read here, in the rare branch,
        //  instead of living in two scalar registers through every trip; null: g.wide_exc)
        const NSK_SCALAR uint32_t *xp = (wexc_slot ? *(const NSK_SCALAR uint32_t *const NSK_SCALAR *)wexc_slot
                                                   : (const NSK_SCALAR uint32_t *)g.wide_exc) + 2 * (size_t)(exc0 + e);
        const uint32_t ex = xp[0], eid = xp[1];
        const uint32_t o = ex & 0xFFu, sh = 8u * (o & 3u) + ((ex >> 8) & 7u);
        if ((uint32_t)lane == (o >> 2)) idx4 = (idx4 & ~(1u << sh)) | (((uint32_t)(uint8_t)g.val[eid] & 1u) << sh);
    }
    uint32_t thr[4], hi[4], out = 0;
    bool tie = false;
#pragma unroll
    for (int i = 0; i < 4; i++) thr[i] = zt[(idx4 >> (8 * i)) & 0xFFu];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        hi[i] = word_of(ra, (uint32_t)i) >> 5;
        out |= (hi[i] > thr[i] ? 1u : 0u) << (8 * i);
        tie = tie || hi[i] == thr[i];
    }
    if (__builtin_expect(__any(tie), 0)) {                                  // 2^-27 per draw: the low 26 bits decide
        const u32x4 rb = philox4x32(k0, k1, blk, 3u, s0, s1);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (hi[i] == thr[i]) {
                const uint32_t lo = g.ztab[zoff + ((idx4 >> (8 * i)) & 0xFFu)].y;
                out = (out & ~(1u << (8 * i))) | (((word_of(rb, (uint32_t)i) >> 6) > lo ? 1u : 0u) << (8 * i));
            }
    }
    {
    // packed mode: ONE store -- per byte (tally << 1) + 2 * value | value; no carry between the bytes while a tally stays
    // below 127 (the host unpacks before)
    // (write-through and non-temporal stores were measured too, tools/sessions/r6_s07.sh: 21.0 / 19.7 against 19.9 us per
    // 10M-grid sweep -- plain stores stay)
    uint32_t *const vq = (uint32_t *)((char *)g.val + ((uint32_t)p0 + 4u * (uint32_t)lane));
    if (MODE == 2) *vq = (t.tally & 0xFEFEFEFEu) + 3u * out;
    else {
        *vq = out;
        if (MODE == 0) *(uint32_t *)((char *)g.cnt_pos + ((uint32_t)p0 + 4u * (uint32_t)lane)) = t.tally + out;   // (four byte tallies: folded before one reaches 255)
    }
    }
}
template <int NCH>
__device__ __forceinline__ void tab_quad_wide(const DevGraph<signed char> &g, const SegEntry &en, int p0,
                                              const uint32_t (&wd)[NSK_WIDE_STRIDE(NCH)], const uint32_t *zt, int lane, int burnin,
                                              uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1) {
    uint32_t base[4 * NCH];
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) base[j] = wd[j];
    WideTrip<NCH> t;
    if (burnin) {
        wide_issue<NCH, 1>(g, p0, base, lane, t);
        wide_finish<NCH, 1>(g, en.zoff, p0, wd[4 * NCH], wd[4 * NCH + 1], wd[4 * NCH + 2], t, zt, lane, k0, k1, s0, s1);
    } else {
        wide_issue<NCH, 0>(g, p0, base, lane, t);
        wide_finish<NCH, 0>(g, en.zoff, p0, wd[4 * NCH], wd[4 * NCH + 1], wd[4 * NCH + 2], t, zt, lane, k0, k1, s0, s1);
    }
}
// the block's copy of the launch's draw tables (thresholds' top 27 bits), segment s at zt + (s << NSK_ZT_BITS(NCH)).
// Two halves so that a kernel can put its first trip's requests between them: tab_fill_request issues the loads
// (per thread up to NSK_ZT_PER_THREAD entries; the segment entries come from the kernel arguments by scalar loads and a
// select chain -- a per-lane index into them would be a vector load of its own in front of the table's), tab_fill_land
// stores them to LDS and closes with the block's barrier.
#define NSK_ZT_PER_THREAD(NCH) (((NSK_SEG_MAX << NSK_ZT_BITS(NCH)) + NSK_BLOCK - 1) / NSK_BLOCK)
template <int NCH>
__device__ __forceinline__ void tab_fill_request(const uint4 *ztab, const SegTable &tab, uint32_t (&v)[NSK_ZT_PER_THREAD(NCH)]) {
    const int per = 1 << NSK_ZT_BITS(NCH);
#pragma unroll
    for (int r = 0; r < NSK_ZT_PER_THREAD(NCH); r++) {
        const int e = (int)threadIdx.x + r * NSK_BLOCK;
        const int sI = e >> NSK_ZT_BITS(NCH), i = e & (per - 1);
        uint32_t zoff = 0u, zmask = 0u;
        bool live = false;
#pragma unroll
        for (int q = 0; q < NSK_SEG_MAX; q++)
            if (sI == q) { zoff = tab.e[q].zoff; zmask = tab.e[q].zmask_ev & 0xFFu; live = q < tab.n; }
        v[r] = 0u;
        if (live && (uint32_t)i <= zmask) v[r] = ztab[zoff + (uint32_t)i].x;
    }
}
template <int NCH>
__device__ __forceinline__ void tab_fill_land(uint32_t *zt, const uint32_t (&v)[NSK_ZT_PER_THREAD(NCH)]) {
#pragma unroll
    for (int r = 0; r < NSK_ZT_PER_THREAD(NCH); r++) {
        const int e = (int)threadIdx.x + r * NSK_BLOCK;
        if (e < (NSK_SEG_MAX << NSK_ZT_BITS(NCH))) zt[e] = v[r];
    }
    __syncthreads();
}
template <int NCH>
__device__ __forceinline__ void tab_fill_lds(const uint4 *ztab, const SegTable &tab, uint32_t *zt) {
    uint32_t v[NSK_ZT_PER_THREAD(NCH)];
    tab_fill_request<NCH>(ztab, tab, v);
    tab_fill_land<NCH>(zt, v);
}

// The quads of an XCD's eighth are dealt to its waves in whole rounds -- a wave's trip is a quad --; what is
// left after the last whole round (fewer quads than waves) is dealt as tile PAIRS, two waves to a quad (each
// evaluates the quad's block): the closing trip of a launch is then half as long, and a launch with fewer
// quads than waves (small grids: one wave lifetime long) runs entirely in pairs.
template <typename VT, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_seg_tab(DevGraph<VT> g, SegTable tab, int burnin,
                                                             uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1,
                                                             const unsigned long long *sweep_base, uint32_t sweep_off) {
    if (sweep_base) {             // a captured launch (hipGraph): sweep index, key and shard tag live in device memory
        const NSK_SCALAR unsigned long long *cb = (const NSK_SCALAR unsigned long long *)sweep_base;
        const unsigned long long sw = cb[0] + sweep_off, key = cb[2];
        s0 = (uint32_t)sw;
        s1 = (uint32_t)(sw >> 32) ^ (uint32_t)cb[3];
        k0 = (uint32_t)key;
        k1 = (uint32_t)(key >> 32);
    }
    const int lane = (int)(threadIdx.x & 63);
    const int nquads = tab.ntiles >> 2;                                 // virtual tiles: a multiple of 4
    const int per = (nquads + 7) >> 3;                                  // quads per XCD
    const int xcd = (int)(blockIdx.x & 7);
    const int wx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    const int wpx = (int)(gridDim.x >> 3) * (NSK_BLOCK / 64);           // waves per XCD
    const int q0 = min(nquads, xcd * per), q1 = min(nquads, (xcd + 1) * per);
    const int qr = q0 + ((q1 - q0) / wpx) * wpx;                        // first quad dealt as pairs
    // the segment of the last located quad stays in scalar registers: most launches have one
    int c_lo = 0, c_hi = -1;
    SegEntry en = tab.e[0];
    // units: quads [q0, qr) one per trip, then the pairs of quads [qr, q1)
    for (int U = wx; ; U += wpx) {
        int Q, h = -1;
        if (q0 + U < qr) Q = q0 + U;
        else {
            const int u = U - (qr - q0);                                // pair unit of the remainder
            if (u >= 2 * (q1 - qr)) break;
            Q = qr + (u >> 1);
            h = u & 1;
        }
        if (4 * Q < c_lo || 4 * Q >= c_hi) {                            // wave-uniform, rare
            const int sidx = seg_of_tile(tab, 4 * Q);
            en = tab.e[sidx];
            c_lo = en.tile_start;
            c_hi = sidx + 1 < NSK_SEG_MAX ? tab.e[sidx + 1].tile_start : tab.ntiles;
        }
        const int lead = (int)(en.ntiles_lead >> 30);
        const int t0q = 4 * Q - en.tile_start - lead;                   // segment tile of the quad's first tile
        // (a wide quad -- nsk_compile.h seg_wide -- in a launch that is not the wide kernel's, most of its quads being
        // of the other kind: tile by tile with the wide scheme's words)
        bool ws = false;
        if (sizeof(VT) == 1 && en.wide_off != NSK_NO_STREAM)
            ws = ((const NSK_SCALAR uint32_t *)(g.seg_wide + en.wide_off))[(size_t)(Q - (en.tile_start >> 2)) * NSK_WIDE_STRIDE(NCH)] != 0xFFFFFFFFu;
        // the quad's block: en.pos0 - 64 lead is a multiple of 256, so (pos >> 8, lane) names it
        const uint32_t qb = quad_block((uint32_t)(en.pos0 + t0q * 64 + lane));
        const u32x4 ra = philox4x32(k0, k1, qb, 2u, s0, s1);
        u32x4 rb = {0u, 0u, 0u, 0u};
        bool have_b = false;
        if (h >= 0) {                                                   // wave-uniform
            tab_tiles<VT, NCH, 2>(g, en, t0q + 2 * h, 2 * h, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
        } else if (NSK_TAB_BATCH == 4) {
            tab_tiles<VT, NCH, 4>(g, en, t0q, 0, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
        } else {
            tab_tiles<VT, NCH, 2>(g, en, t0q, 0, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
            tab_tiles<VT, NCH, 2>(g, en, t0q + 2, 2, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
        }
    }
}

// ---- the table launch of a class whose quads are (mostly) wide ones (SegTable.wide; int8 values) ----
// Resident grid, XCD x walks the x-th eighth of the launch's quads, a wave's trip is a quad: value loads (the quad's
// descriptor -- slot bases, exception list, slot mask -- was requested one trip ahead by a scalar load) -> Philox block
// while they fly -> pack -> four look-ups in the wave's LDS copy of the segment's draw table -> compares -> one or two
// dword stores.  A quad that is not wide (a class end, a row of mixed border cells) is sampled tile by tile by the
// round-4 routine, one tile at a time.
// MODE 0: the tally in cnt_pos; 1: burn-in; 2: PACKED -- the sweep's tally lives in the value bytes themselves (bit 0 the
// value, bits 1-7 the tallied sweeps it was 1 since the last k_unpack_tally, at most 127 apart): a trip reads the quad's
// own bytes instead of a tally array and leaves with ONE store.  Every reader of values masks bit 0 while the mode is on:
// it is on only for whole-graph handles whose every launch is this kernel's (nsk_gibbs.hip pack_now).
// WHAT A WAVE'S LIFE IS MADE OF (per-wave s_memtime, tools/timing_tabw.py, 10M grid, r6_s10.sh): with the segment table in
// the kernel arguments and a block-wide LDS table behind a barrier, 56 % of it lay in FRONT of the first trip -- five
// dependent rounds of loads (kernel arguments -> segment entry -> table entries / descriptor -> ...) at ~1 us each: the
// scalar cache and the L2s start every launch cold.  Hence:
// * the arguments the first trip needs are the kernel's first 16 dwords and the translation unit is compiled with
//   kernarg preloading (Makefile: -amdgpu-kernarg-preload-count=16): they are in scalar registers when the wave starts;
//   the first segment's entry is among them (the largest segment comes first), other segments load theirs later;
// * the draw table is copied per WAVE and segment (16 or 256 thresholds, lanes = entries): no block barrier, and the
//   copy's loads go out with the first descriptor request;
// * everything else in the arguments (the other segment entries, the fall-back's arrays) is touched behind the first
//   trip's requests.
#ifndef NSK_TABW_ATTR
#define NSK_TABW_ATTR
#endif
#ifdef NSK_ABL_TIMING       // instrumented build (tools/build_variant.sh TIMING -DNSK_ABL_TIMING; tools/timing_tabw.py)
static __device__ unsigned long long nsk_dbg[4 * 65536];
#endif
// The hot arguments: 14 dwords (16 user scalar registers less the kernarg pointer's two).  The three other arrays are
// given as 256-byte units from `val` (device allocations are 256-byte aligned and lie within 512 GB of each other).
// A scalar that a (conditional) scalar load inside the trip loop produced, re-defined by an ALU move: the wait for the
// load sits HERE.  Without it every later use inside the loop may find the register "possibly pending" (the compiler's
// wait counters merge at the loop header; scalar loads return out of order, so the only wait is lgkmcnt(0)) -- and that
// wait also drains the descriptor request a trip ahead right after it was issued.
__device__ __forceinline__ uint32_t nsk_settled(uint32_t x) { uint32_t y; asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x)); return y; }
template <typename T> __device__ __forceinline__ T *nsk_settled_ptr(T *p) {
    const unsigned long long v = (unsigned long long)p;
    return (T *)(((unsigned long long)nsk_settled((uint32_t)(v >> 32)) << 32) | nsk_settled((uint32_t)v));
}
#define NSK_TABW_HOT signed char *val, int d_cnt, int d_wide, int d_ztab, int ts1, uint32_t ntl0, int pos00, uint32_t woff0,          \
                     uint32_t zoff0, uint32_t ntiles_nseg, uint32_t zmask0_wpx, const unsigned long long *sweep_base
#define NSK_TABW_REST_MAX 64        // quads of a launch that are not wide ones and get a workgroup of their own (below)
struct TabwCold {                   // ... and the rest, read through a laundered pointer BEHIND the first trip's requests
    uint32_t k0, k1, s0, s1;        //     (the compiler hoists every load of a kernel argument to the kernel's entry)
    uint32_t sweep_off, pad_;
    DevGraph<signed char> g;
    SegTable tab;
    // The launch's quads that are NOT wide ones (a grid's border columns, class ends): virtual quad | bit 31: its positions
    // draw from the wide scheme all the same.  Each is sampled by a workgroup of its own IN FRONT of the grid (a wave per
    // tile, blockIdx < zmask0_wpx >> 24), not by the wave whose turn it would be: sampled in line, tile after tile, these
    // few quads -- 11 of a 10M-grid launch's 20 011 -- kept their waves busy long after the others had finished: 1.6 of the
    // launch's 8.7 us (tools/sessions/r6_s21.sh, r6_s22.sh: the fall-back compiled in but never run 7.1 us, left out 7.1).
    uint32_t nrest, rest[NSK_TABW_REST_MAX];
};
#define NSK_TABW_COLD_OFFSET 56     // byte offset of the TabwCold argument in the kernarg segment (14 dwords in front, 8-aligned)
// One of the launch's quads that are not wide ones (TabwCold.rest), a wave per tile, by the round-4 routine; nothing here
// is in a hurry: the waves of these workgroups live about as long as the others'.
template <int NCH, int MODE>
__device__ __forceinline__ void tabw_rest_quad(const char NSK_SCALAR *ka, const unsigned long long *sweep_base) {
    const TabwCold NSK_SCALAR *cold = (const TabwCold NSK_SCALAR *)(ka + NSK_TABW_COLD_OFFSET);
    if (blockIdx.x >= cold->nrest) return;
    const int lane = (int)(threadIdx.x & 63);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t ent = cold->rest[blockIdx.x];
    const int Q = (int)(ent & 0x7FFFFFFFu);
    const bool flagged = (ent >> 31) != 0u;
    const SegTable NSK_SCALAR *tb = &cold->tab;
    int sidx = 0;
    for (int i = 1; i < NSK_SEG_MAX && 4 * Q >= tb->e[i].tile_start; i++) sidx = i;      // (tile_start = ntiles beyond the last segment)
    SegEntry en;
    {
        const NSK_SCALAR uint32_t *ep = (const NSK_SCALAR uint32_t *)(ka + NSK_TABW_COLD_OFFSET + offsetof(TabwCold, tab) +
                                                                      offsetof(SegTable, e) + sizeof(SegEntry) * (size_t)sidx);
        uint32_t ew[sizeof(SegEntry) / 4];
#pragma unroll
        for (int j = 0; j < (int)(sizeof(SegEntry) / 4); j++) ew[j] = ep[j];
        __builtin_memcpy(&en, ew, sizeof(SegEntry));
    }
    uint32_t k0 = cold->k0, k1 = cold->k1, s0 = cold->s0, s1 = cold->s1;
    if (sweep_base) {                 // a captured launch (hipGraph): sweep index, key and shard tag live in device memory
        const NSK_SCALAR unsigned long long *cb = (const NSK_SCALAR unsigned long long *)sweep_base;
        const unsigned long long sw = cb[0] + cold->sweep_off, key = cb[2];
        s0 = (uint32_t)sw;
        s1 = (uint32_t)(sw >> 32) ^ (uint32_t)cb[3];
        k0 = (uint32_t)key;
        k1 = (uint32_t)(key >> 32);
    }
    DevGraph<signed char> gf;
    gf.val = cold->g.val; gf.cnt_pos = cold->g.cnt_pos; gf.ztab = cold->g.ztab; gf.seg_wide = cold->g.seg_wide;
    gf.adj = cold->g.adj; gf.seg_aff = cold->g.seg_aff; gf.sink = cold->g.sink;
    const int lead = (int)(en.ntiles_lead >> 30);
    const int t0q = 4 * Q - en.tile_start - lead;                       // segment tile of the quad's first tile
    const int p0 = en.pos0 + t0q * 64;                                  // the quad's first position (a multiple of 256)
    const uint32_t qb = quad_block((uint32_t)(p0 + lane));
    const u32x4 ra = philox4x32(k0, k1, qb, 2u, s0, s1);
    u32x4 rb = {0u, 0u, 0u, 0u};
    bool have_b = false;
    tab_tiles<signed char, NCH, 1, MODE == 2>(gf, en, t0q + wv, wv, lane, MODE == 1 ? 1 : 0, ra, rb, have_b, qb, k0, k1, s0, s1, flagged);
}
template <int NCH, int MODE>
__global__ __launch_bounds__(NSK_BLOCK) NSK_TABW_ATTR void k_gibbs_seg_tabw(NSK_TABW_HOT, TabwCold cold_unused) {
    const int lane = (int)(threadIdx.x & 63);
#ifdef NSK_ABL_TIMING       // (instrumented build, tools/timing_tabw.py: per wave entry / first requests out / first trip done / exit)
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long dbg_t1 = 0, dbg_t2 = 0;
    int dbg_trips = 0;
#endif
    constexpr int ST = NSK_WIDE_STRIDE(NCH), ZN = 1 << NSK_ZT_BITS(NCH), ZR = (ZN + 63) / 64;
    __shared__ uint32_t zt_all[(NSK_BLOCK / 64) * ZN];                  // one table per wave
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *zt = zt_all + wv * ZN;
    uint8_t *cnt_pos = (uint8_t *)val + (long long)d_cnt * 256;
    const uint32_t *seg_wide = (const uint32_t *)((const char *)val + (long long)d_wide * 256);
    const uint4 *ztab = (const uint4 *)((const char *)val + (long long)d_ztab * 256);
    const int ntiles = (int)(ntiles_nseg & 0x0FFFFFFFu);          // (bits 28-31: segments - 1; tile_start = ntiles beyond them)
    const uint32_t zmask0 = zmask0_wpx & 0xFFu;
    const int wpx = (int)((zmask0_wpx >> 8) & 0xFFFFu);                 // waves per XCD
    const int nfront = (int)(zmask0_wpx >> 24);                         // workgroups in front that sample the quads that are not wide (a multiple of 8)
    if ((int)blockIdx.x < nfront) {                                     // (block-uniform)
        tabw_rest_quad<NCH, MODE>((const char NSK_SCALAR *)__builtin_amdgcn_kernarg_segment_ptr(), sweep_base);
        return;
    }
    const int bid = (int)blockIdx.x - nfront;
    const int nquads = ntiles >> 2;                                     // virtual tiles: a multiple of 4
    const int per = (nquads + 7) >> 3;                                  // quads per XCD
    const int xcd = bid & 7;
    const int wx = __builtin_amdgcn_readfirstlane((bid >> 3) * (NSK_BLOCK / 64)) + wv;
    const int q0 = min(nquads, xcd * per), q1 = min(nquads, (xcd + 1) * per);
    // the cold arguments: nothing of them is read before the first trip's requests are out (`keyed`)
    const char NSK_SCALAR *ka = (const char NSK_SCALAR *)__builtin_amdgcn_kernarg_segment_ptr();
    const TabwCold NSK_SCALAR *cold = nullptr;
    uint32_t k0 = 0, k1 = 0, s0 = 0, s1 = 0;
    DevGraph<signed char> gh;
    gh.val = val; gh.cnt_pos = cnt_pos; gh.ztab = ztab; gh.seg_wide = seg_wide;
    PhiloxKeys pk;
    bool keyed = false;
    (void)cold_unused;
    // The wave's quads: Q = q0 + wx, + wpx, ... < q1, through the launch's segments in order.  The first segment's entry
    // is in the preloaded arguments; a wave whose quad lies in a later one reads that entry from the kernel arguments
    // (every segment is whole quads: no quad straddles two).  The trips inside a segment are the inner loop: the scalar
    // unit is shared by the CU's waves, one instruction per cycle, so what a trip does not need to recompute is hoisted.
    int sidx = 0, tstart = 0, tend = ts1, pos0s = pos00;
    uint32_t ntl = ntl0, woff = woff0, zoff = zoff0, zmask = zmask0;
    for (int Q = q0 + wx; Q < q1;) {
        if (4 * Q >= tend) {                                            // (wave-uniform) on to the quad's segment
            if (!cold) { asm volatile("" : "+s"(ka)); cold = (const TabwCold NSK_SCALAR *)(ka + NSK_TABW_COLD_OFFSET); }
            const SegTable NSK_SCALAR *tb = &cold->tab;
            do {
                sidx++;
                tend = sidx + 1 < NSK_SEG_MAX ? tb->e[sidx + 1].tile_start : ntiles;
            } while (4 * Q >= tend);
            tstart = (int)nsk_settled((uint32_t)tb->e[sidx].tile_start); pos0s = (int)nsk_settled((uint32_t)tb->e[sidx].pos0);
            ntl = nsk_settled(tb->e[sidx].ntiles_lead); woff = nsk_settled(tb->e[sidx].wide_off);
            zoff = nsk_settled(tb->e[sidx].zoff); zmask = nsk_settled(tb->e[sidx].zmask_ev) & 0xFFu;
            tend = (int)nsk_settled((uint32_t)tend);
        }
        const int qs = tstart >> 2;                                     // the segment's first quad of the launch
        const int Qe = min(q1, tend >> 2);                              // ... and the end of its quads in this XCD's share
        const int lead = (int)(ntl >> 30), nt = (int)(ntl & 0x3FFFFFFFu);
        // quads [qin_lo, qin_lo + qin_n) of the launch lie wholly inside the run (all but a first one with dead lead tiles
        // and a last one with fewer than four tiles)
        const int qin_lo = qs + (lead ? 1 : 0);
        const uint32_t qin_n = (uint32_t)(qs + ((lead + nt) >> 2) - qin_lo);
        const bool hasw = woff != NSK_NO_STREAM;
        // descriptor of quad Q (!hasw: words of the array's front, not looked at), its address advancing with the trips
        const NSK_SCALAR uint32_t *wq = (const NSK_SCALAR uint32_t *)(seg_wide + (hasw ? woff : 0u)) + (hasw ? (size_t)(Q - qs) * ST : 0);
        const size_t wstep = hasw ? (size_t)wpx * ST : 0;
        int p0 = pos0s + (4 * Q - tstart - lead) * 64;                  // the quad's first position
        uint32_t cur[ST];
#pragma unroll
        for (int j = 0; j < ST; j++) cur[j] = wq[j];
        // the wave's copy of the segment's thresholds (their top 27 bits), lanes = entries: requested here, stored to LDS
        // behind the first trip's value requests
        uint32_t ztv[ZR];
#pragma unroll
        for (int r = 0; r < ZR; r++) {
            const uint32_t e = (uint32_t)lane + 64u * (uint32_t)r;
            ztv[r] = e <= zmask ? ztab[zoff + e].x : 0u;
        }
        bool land = true;
        for (; Q < Qe; Q += wpx, p0 += 256 * wpx) {
#ifdef NSK_ABL_TIMING
            if (dbg_trips == 1) { __builtin_amdgcn_s_waitcnt(0); dbg_t2 = __builtin_amdgcn_s_memtime(); }
            dbg_trips++;
#endif
            // (the descriptor's registers are free for the next one as soon as the bases are in the loads' addresses)
            const bool flagged = hasw && cur[0] != 0xFFFFFFFFu, wide = flagged && (uint32_t)(Q - qin_lo) < qin_n;
            const uint32_t exc0 = cur[4 * NCH], nexc = cur[4 * NCH + 1], smask = cur[4 * NCH + 2];
            asm volatile("" :: "s"(cur[4 * NCH + 3])); // (the descriptor's spare word stays live while the request flies: its
                                                       //  register handed to a temporary means a wait for the whole request)
            WideTrip<NCH> tc;
            if (wide) {
                uint32_t bc[4 * NCH];
#pragma unroll
                for (int j = 0; j < 4 * NCH; j++) bc[j] = cur[j];
                wide_issue<NCH, MODE>(gh, p0, bc, lane, tc);
            }
            // the next trip's descriptor, a scalar round trip ahead (always a load, into the registers the bases just
            // left: past the segment's quads this quad's again)
            if (Q + wpx < Qe) wq += wstep;
#pragma unroll
            for (int j = 0; j < ST; j++) cur[j] = wq[j];
            if (land) {
                land = false;
                if (!keyed) {             // the wave's first requests are out: now the cold arguments
                    keyed = true;
                    if (!cold) { asm volatile("" : "+s"(ka)); cold = (const TabwCold NSK_SCALAR *)(ka + NSK_TABW_COLD_OFFSET); }
                    k0 = cold->k0; k1 = cold->k1; s0 = cold->s0; s1 = cold->s1;
                    if (sweep_base) {     // a captured launch (hipGraph): sweep index, key and shard tag live in device memory
                        const NSK_SCALAR unsigned long long *cb = (const NSK_SCALAR unsigned long long *)sweep_base;
                        const unsigned long long sw = cb[0] + cold->sweep_off, key = cb[2];
                        s0 = (uint32_t)sw;
                        s1 = (uint32_t)(sw >> 32) ^ (uint32_t)cb[3];
                        k0 = (uint32_t)key;
                        k1 = (uint32_t)(key >> 32);
                    }
                    k0 = nsk_settled(k0); k1 = nsk_settled(k1); s0 = nsk_settled(s0); s1 = nsk_settled(s1);
                    pk = philox_round_keys(k0, k1);
                }
#pragma unroll
                for (int r = 0; r < ZR; r++)
                    if (lane + 64 * r < ZN) zt[lane + 64 * r] = ztv[r];
                asm volatile("" ::: "memory");                          // (LDS executes a wave's accesses in order: no barrier)
#ifdef NSK_ABL_TIMING
                if (!dbg_t1) dbg_t1 = __builtin_amdgcn_s_memtime();
#endif
            }
            if (wide) {
                wide_finish<NCH, MODE>(gh, zoff, p0, exc0, nexc, smask, tc, zt, lane, k0, k1, s0, s1,
                                       ka + NSK_TABW_COLD_OFFSET + offsetof(TabwCold, g) + offsetof(DevGraph<signed char>, wide_exc), &pk);
                continue;
            }
            // not a wide quad (a class end, mixed border cells): a workgroup in front has it -- or, in a launch with more
            // such quads than those take, tile by tile here.  Its block: the run's first position less its lead tiles is
            // a multiple of 256, so (pos >> 8, lane) names it
            if (nfront) continue;
            SegEntry en;
            {
                const NSK_SCALAR uint32_t *ep = (const NSK_SCALAR uint32_t *)(ka + NSK_TABW_COLD_OFFSET + offsetof(TabwCold, tab) +
                                                                              offsetof(SegTable, e) + sizeof(SegEntry) * (size_t)sidx);
                uint32_t ew[sizeof(SegEntry) / 4];
#pragma unroll
                for (int j = 0; j < (int)(sizeof(SegEntry) / 4); j++) ew[j] = ep[j];
                __builtin_memcpy(&en, ew, sizeof(SegEntry));
            }
            DevGraph<signed char> gf = gh;                              // (the fall-back's arrays: read here, not kept)
            gf.adj = cold->g.adj; gf.seg_aff = cold->g.seg_aff; gf.sink = cold->g.sink;
            const int t0q = 4 * Q - tstart - lead;
            const uint32_t qb = quad_block((uint32_t)(p0 + lane));
            const u32x4 ra = philox4x32(k0, k1, qb, 2u, s0, s1);
            u32x4 rb = {0u, 0u, 0u, 0u};
            bool have_b = false;
#pragma unroll 1
            for (int t = 0; t < 4; t++)
                tab_tiles<signed char, NCH, 1, MODE == 2>(gf, en, t0q + t, t, lane, MODE == 1 ? 1 : 0, ra, rb, have_b, qb, k0, k1, s0, s1, flagged);
        }
    }
#ifdef NSK_ABL_TIMING
    {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long dbg_t3 = __builtin_amdgcn_s_memtime();
        const int slot = (int)((blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6)) & 16383u);
        if (lane == 0) {
            nsk_dbg[4 * slot] = dbg_t0; nsk_dbg[4 * slot + 1] = dbg_t1; nsk_dbg[4 * slot + 2] = dbg_t2;
            nsk_dbg[4 * slot + 3] = dbg_t3 | ((unsigned long long)dbg_trips << 56);
        }
    }
#endif
}

// The fused-exchange flavour of tab_tiles / k_gibbs_seg_tab below (P2P = true is the only instantiation; buffer
// addressing: scalar base + lane offset, a dead tile's stores dropped by the bounds check).  The single-GPU kernel keeps
// the round-4 body above: the same restructuring there cost 1.5-2.4 us per 10M-grid class (tools/sessions/r5_s03.sh ..
// r5_s05.sh: buffer loads + stores 14.3 us, flat loads 14.1, flat stores 14.15, both flat in this structure 13.85,
// round-4 body 12.0).
// NT consecutive tiles of one quad, first tile = segment tile `t0` (word `w0` of the quad's blocks): every
// load of the NT tiles is requested before the first draw, the draws' stores come last
template <typename VT, int NCH, int NT, bool P2P>
__device__ __forceinline__ void tab_tiles_x(const DevGraph<VT> &g, const SegEntry &en, int t0, int w0, int lane, int burnin,
                                          const u32x4 &ra, u32x4 &rb, bool &have_b, uint32_t qb,
                                          uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1, const TabP2P &px, uint32_t ptag, bool ws) {
    const int nt = (int)(en.ntiles_lead & 0x3FFFFFFFu);
    const bool border = P2P && en.push_off != NSK_NO_STREAM;           // (wave-uniform)
    nsk_rsrc rg = nsk_make_rsrc(g.val);
    if (P2P) {
        // ghosts: this rank's receive block of the previous exchange (tag - 1), read with system-coherent loads; a
        // border wave first waits (bounded) for the peers' flags of that exchange -- relaxed polls: what the flag
        // guards is read past the caches anyway
        rg = nsk_make_rsrc((const VT *)((const char *)px.mine + nsk_p2p_recv_off_(px.world)) + (size_t)((ptag - 1u) & 1u) * 2 * (size_t)px.nrecv);
        if (border && px.wait) {
            if (lane == 0) {
                const unsigned int *flags = (const unsigned int *)px.mine + (size_t)((ptag - 1u) & 1u) * 2 * (size_t)px.world;
                const unsigned long long tw0 = wall_clock64();
                bool ok = true;
                for (int q = 0; q < px.world && ok; q++) {
                    if (!((px.peer_mask >> q) & 1u)) continue;
                    while (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ptag - 1u) {
                        if (wall_clock64() - tw0 > px.timeout_ticks) { ok = false; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (!ok) (void)__hip_atomic_fetch_or(px.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_wave_barrier();               // (the lanes' ghost loads stay behind lane 0's wait)
        }
    }
    const bool haff = en.aff_off != NSK_NO_STREAM;                      // implicit adjacency (nsk_compile.h seg_aff)
    const uint32_t zoff = en.zoff, zmask = en.zmask_ev & 0xFFu;
    const nsk_rsrc rv = nsk_make_rsrc(g.val), rc = nsk_make_rsrc(g.cnt_pos);
    bool live[NT];
    int tt[NT];
    uint32_t tally[NT];
    uint32_t id[NT][4 * NCH], ab[NT][4 * NCH];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int t = t0 + k;
        live[k] = t >= 0 && t < nt;                                     // wave-uniform
        tt[k] = live[k] ? t : (t < 0 ? 0 : nt - 1);                     // (a dead tile reads a real one's data)
        ab[k][0] = NSK_NO_STREAM;
    }
    if (haff) {                        // the tiles' slot bases: scalar loads, issued together
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const NSK_SCALAR uint32_t *ap = (const NSK_SCALAR uint32_t *)(g.seg_aff + en.aff_off + (size_t)tt[k] * NCH);
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) ab[k][j] = ap[j];
        }
    }
#pragma unroll
    for (int k = 0; k < NT; k++)
        tally[k] = burnin ? 0u : (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rc, lane, en.pos0 + tt[k] * 64, 0);
#pragma unroll
    for (int k = 0; k < NT; k++) {
        if (ab[k][0] == NSK_NO_STREAM) {                                // wave-uniform: the tile reads its stream
            const uint4 *sp = g.adj + en.adj_off + (size_t)tt[k] * (64 * NCH) + lane;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint4 w = sp[c * 64];
                id[k][4 * c] = w.x; id[k][4 * c + 1] = w.y; id[k][4 * c + 2] = w.z; id[k][4 * c + 3] = w.w;
            }
        }
    }
    // member values -> neighbourhood bits -> table entries (the table kernels run only while every value on
    // the device lies in its domain -- values_regular -- and their members are binary: a value IS its bit).
    // Every load of the NT tiles is issued first (the two tile kinds take different paths: only the requests sit
    // inside the wave-uniform branches, the shifts and ors that use them follow behind all of them -- with the
    // uses inside the branches a tile's loads were waited for before the next tile's were requested: 12.0 ->
    // 15.3 us per 10M-grid class, tools/sessions/r5_s03.sh)
    uint32_t x[NT][4 * NCH];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        if (ab[k][0] != NSK_NO_STREAM) {                 // implicit adjacency: member = base of the slot (scalar) + lane
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) {
                const uint32_t b = ab[k][j];
                if (!P2P) {
                    x[k][j] = nsk_buf_ld<VT>(rv, (uint32_t)lane, b);
                } else {                                 // ... a run of the values or of the ghosts ([ghost_lo, ghost_lo + nrecv))
                    const uint32_t gb = b - px.ghost_lo;
                    if (gb + 63u < px.nrecv && gb < px.nrecv) x[k][j] = nsk_buf_ld_sys<VT>(rg, (uint32_t)lane, gb);       // scalar: all ghosts
                    else if (b + 63u < px.ghost_lo || gb >= px.nrecv) x[k][j] = nsk_buf_ld<VT>(rv, (uint32_t)lane, b);   // scalar: none
                    else {
                        const uint32_t i = b + (uint32_t)lane;
                        x[k][j] = i - px.ghost_lo < px.nrecv ? nsk_buf_ld_sys<VT>(rg, i - px.ghost_lo, 0u) : nsk_buf_ld<VT>(rv, i, 0u);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) {
                const uint32_t i = id[k][j];
                if (!P2P) x[k][j] = nsk_buf_ld<VT>(rv, i, 0u);
                else x[k][j] = i - px.ghost_lo < px.nrecv ? nsk_buf_ld_sys<VT>(rg, i - px.ghost_lo, 0u) : nsk_buf_ld<VT>(rv, i, 0u);
            }
        }
    }
    uint32_t idx[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        idx[k] = 0;
#pragma unroll
        for (int j = 0; j < 4 * NCH; j++) idx[k] |= x[k][j] << j;
    }
    uint2 e[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) e[k] = *(const uint2 *)(g.ztab + zoff + (idx[k] & zmask));
    // draws: the high 27 bits decide unless they tie with the threshold's
    uint32_t hi[NT];
    int nv[NT];
    bool tie = false;
    // (ws, wave-uniform: the quad is a WIDE one -- nsk_compile.h seg_wide -- that this launch samples tile by tile, a
    // shard's run boundary cuts it: its positions keep the wide generator scheme, nsk_device.h wide_word_of_tile)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        hi[k] = (ws ? wide_word_of_tile(ra, (uint32_t)(w0 + k), (uint32_t)lane) : word_of(ra, (uint32_t)(w0 + k))) >> 5;
        nv[k] = hi[k] > e[k].x ? 1 : 0;
        tie = tie || hi[k] == e[k].x;
    }
    if (__builtin_expect(__any(tie), 0)) {
        if (!have_b) { rb = philox4x32(k0, k1, qb, 3u, s0, s1); have_b = true; }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const uint32_t lo = (ws ? wide_word_of_tile(rb, (uint32_t)(w0 + k), (uint32_t)lane) : word_of(rb, (uint32_t)(w0 + k))) >> 6;
            if (hi[k] == e[k].x) nv[k] = lo > e[k].y ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < NT; k++) {
        // a dead tile's stores carry an out-of-range lane offset: dropped by the bounds check (no branch, no select
        // of addresses)
        const uint32_t voff = live[k] ? (uint32_t)lane : NSK_BUF_OOB;
        const uint32_t soff = live[k] ? (uint32_t)(en.pos0 + (t0 + k) * 64) : 0u;
        nsk_buf_st<VT>(rv, voff, soff, nv[k]);
        if (!burnin) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(tally[k] + (uint32_t)nv[k]), rc, (int)voff, (int)soff, 0);
    }
    if (border) {
        // the boundary values of these tiles into their readers' receive blocks (this exchange's parity), written
        // through to memory (system-coherent stores); once they are acknowledged the tiles are counted, and the wave
        // that completes the sweep's last border tile raises the flags
        uint32_t nlive = 0;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            if (!live[k]) continue;                                     // wave-uniform
            nlive++;
            const uint32_t pm = px.push_map[((size_t)en.push_off + (size_t)(t0 + k)) * 64 + lane];
            for (int q = 0; q < px.world; q++) {
                if (!((px.peer_mask >> q) & 1u)) continue;              // (scalar loop: a shard of a grid has two readers)
                const nsk_rsrc rp = nsk_make_rsrc((VT *)((char *)px.peer[q] + nsk_p2p_recv_off_(px.world)) + (size_t)(ptag & 1u) * 2 * (size_t)px.dtotal[q]);
                nsk_buf_st_sys<VT>(rp, (pm != NSK_NO_STREAM && (int)(pm >> 28) == q) ? (pm & 0x0FFFFFFFu) : NSK_BUF_OOB, nv[k]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): the stores above have been acknowledged
        if (lane == 0) {
            const unsigned int old = __hip_atomic_fetch_add(px.counter, nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + nlive == px.border_total) {
                __hip_atomic_store(px.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int q = 0; q < px.world; q++)
                    if ((px.peer_mask >> q) & 1u)
                        __hip_atomic_store((unsigned int *)px.peer[q] + (size_t)(ptag & 1u) * 2 * (size_t)px.world + (size_t)px.me, ptag,
                                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <typename VT, int NCH, bool P2P>
__device__ __forceinline__ void gibbs_seg_tab_body(const DevGraph<VT> &g, const SegTable &tab, int burnin,
                                                   uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1,
                                                   const unsigned long long *sweep_base, uint32_t sweep_off, const TabP2P &px) {
    uint32_t ptag = P2P ? px.tag : 0u;
    if (P2P && px.tag_base) ptag += (uint32_t)((const NSK_SCALAR unsigned long long *)px.tag_base)[1];
    if (sweep_base) {             // a captured launch (hipGraph): sweep index, key and shard tag live in device memory
        const NSK_SCALAR unsigned long long *cb = (const NSK_SCALAR unsigned long long *)sweep_base;
        const unsigned long long sw = cb[0] + sweep_off, key = cb[2];
        s0 = (uint32_t)sw;
        s1 = (uint32_t)(sw >> 32) ^ (uint32_t)cb[3];
        k0 = (uint32_t)key;
        k1 = (uint32_t)(key >> 32);
    }
    const int lane = (int)(threadIdx.x & 63);
    __shared__ uint32_t zt[NSK_SEG_MAX << NSK_ZT_BITS(NCH)];             // wide quads read their thresholds from LDS
    constexpr bool WIDE = sizeof(VT) == 1;
    if (WIDE && tab.wide) tab_fill_lds<NCH>(g.ztab, tab, zt);
    const int nquads = tab.ntiles >> 2;                                 // virtual tiles: a multiple of 4
    const int per = (nquads + 7) >> 3;                                  // quads per XCD
    const int xcd = (int)(blockIdx.x & 7);
    const int wx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    const int wpx = (int)(gridDim.x >> 3) * (NSK_BLOCK / 64);           // waves per XCD
    const int q0 = min(nquads, xcd * per), q1 = min(nquads, (xcd + 1) * per);
    const int qr = q0 + ((q1 - q0) / wpx) * wpx;                        // first quad dealt as pairs
    // the segment of the last located quad stays in scalar registers: most launches have one
    int c_lo = 0, c_hi = -1, sidx = 0;
    SegEntry en = tab.e[0];
    // units: quads [q0, qr) one per trip, then the pairs of quads [qr, q1)
    for (int U = wx; ; U += wpx) {
        int Q, h = -1;
        if (q0 + U < qr) Q = q0 + U;
        else {
            const int u = U - (qr - q0);                                // pair unit of the remainder
            if (u >= 2 * (q1 - qr)) break;
            Q = qr + (u >> 1);
            h = u & 1;
        }
        if (4 * Q < c_lo || 4 * Q >= c_hi) {                            // wave-uniform, rare
            sidx = seg_of_tile(tab, 4 * Q);
            en = tab.e[sidx];
            c_lo = en.tile_start;
            c_hi = sidx + 1 < NSK_SEG_MAX ? tab.e[sidx + 1].tile_start : tab.ntiles;
        }
        const int lead = (int)(en.ntiles_lead >> 30);
        const int t0q = 4 * Q - en.tile_start - lead;                   // segment tile of the quad's first tile
        // a wide quad (nsk_compile.h seg_wide) of an interior run goes four positions to a lane when the run holds all of
        // it (of a quad dealt as two pairs the first wave takes it whole); one that a run boundary cuts, or a border
        // run's, is sampled tile by tile with the wide scheme's words (ws)
        bool ws = false;
        if constexpr (WIDE) {
            if (en.wide_off != NSK_NO_STREAM) {
                const NSK_SCALAR uint32_t *wp = (const NSK_SCALAR uint32_t *)(g.seg_wide + en.wide_off +
                                                                                (size_t)(Q - (en.tile_start >> 2)) * NSK_WIDE_STRIDE(NCH));
                uint32_t wd[NSK_WIDE_STRIDE(NCH)];
#pragma unroll
                for (int j = 0; j < NSK_WIDE_STRIDE(NCH); j++) wd[j] = wp[j];
                ws = wd[0] != 0xFFFFFFFFu;
                if (ws && en.push_off == NSK_NO_STREAM && t0q >= 0 && t0q + 4 <= (int)(en.ntiles_lead & 0x3FFFFFFFu)) {
                    if (h <= 0) tab_quad_wide<NCH>(g, en, en.pos0 + t0q * 64, wd, zt + (sidx << NSK_ZT_BITS(NCH)), lane, burnin, k0, k1, s0, s1);
                    continue;
                }
            }
        }
        // the quad's block: en.pos0 - 64 lead is a multiple of 256, so (pos >> 8, lane) names it
        const uint32_t qb = quad_block((uint32_t)(en.pos0 + t0q * 64 + lane));
        const u32x4 ra = philox4x32(k0, k1, qb, 2u, s0, s1);
        u32x4 rb = {0u, 0u, 0u, 0u};
        bool have_b = false;
        // interior runs (no tile of the segment touches the boundary: none reads a ghost, none pushes) take the
        // single-GPU body as it is -- only the border runs, a percent of a shard's tiles, pay for the other one
        if (en.push_off == NSK_NO_STREAM) {                             // wave-uniform
            if (h >= 0) tab_tiles<VT, NCH, 2>(g, en, t0q + 2 * h, 2 * h, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
            else tab_tiles<VT, NCH, 4>(g, en, t0q, 0, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, ws);
        } else if (h >= 0) {
            tab_tiles_x<VT, NCH, 2, P2P>(g, en, t0q + 2 * h, 2 * h, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, px, ptag, ws);
        } else {
            tab_tiles_x<VT, NCH, 4, P2P>(g, en, t0q, 0, lane, burnin, ra, rb, have_b, qb, k0, k1, s0, s1, px, ptag, ws);
        }
    }
}
// the same for a shard that exchanges its boundary inside the launch (TabP2P)
template <typename VT, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_seg_tab_p2p(DevGraph<VT> g, SegTable tab, int burnin,
                                                                 uint32_t k0, uint32_t k1, uint32_t s0, uint32_t s1,
                                                                 const unsigned long long *sweep_base, uint32_t sweep_off, TabP2P px) {
    gibbs_seg_tab_body<VT, NCH, true>(g, tab, burnin, k0, k1, s0, s1, sweep_base, sweep_off, px);
}

template <typename VT, int KIND, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_seg(DevGraph<VT> g, SegTable tab, int nblocks,
                                                         int burnin, uint32_t k0, uint32_t k1,
                                                         uint32_t s0, uint32_t s1) {
    const int lb = xcd_logical_block((int)blockIdx.x, nblocks);
    if (lb < 0) return;
    const int lane = (int)(threadIdx.x & 63);
    const int T = __builtin_amdgcn_readfirstlane(lb * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (T >= tab.ntiles) return;
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < NSK_SEG_MAX; i++) sidx += (i < tab.n && T >= tab.e[i].tile_start) ? 1 : 0;
    const SegEntry en = tab.e[sidx];
    const int t = T - en.tile_start;
    const uint32_t prog = en.prog;
    const int p = en.pos0 + t * 64 + lane;
    const int v = g.p_vid[p];                             // -1: padding lane at a class end
    const uint8_t tally = burnin ? (uint8_t)0 : g.cnt_pos[p];
    const uint4 *sp = g.adj + en.adj_off + (size_t)t * (64 * NCH) + lane;
    const NSK_SCALAR double *tw = (const NSK_SCALAR double *)(g.prog_w + 2 * (size_t)prog);
    const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    uint4 q[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) q[c] = sp[c * 64];     // (the non-temporal hint costs 10 % here)
    int x[4 * NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        x[4 * c] = (int)g.val[q[c].x]; x[4 * c + 1] = (int)g.val[q[c].y];
        x[4 * c + 2] = (int)g.val[q[c].z]; x[4 * c + 3] = (int)g.val[q[c].w];
    }
    double p0 = 0.0, p1 = 0.0;
    SlotState st = {0, true, false, true};
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) {
        const double thi = tw[2 * j], tlo = tw[2 * j + 1];
        if (KIND) pair_step<KIND>(thi, tlo, x[j], p0, p1);
        else slot_step(st, pp[j], thi, tlo, x[j], p0, p1);
    }
    // a segment with a draw table keeps its positions' quad scheme when the exp path runs instead of the
    // table kernel (a value outside its domain has been uploaded): wave-uniform
    // ... and inside it the wide scheme where the quad is a wide one (nsk_compile.h seg_wide)
    bool wq = false;
    if (en.wide_off != NSK_NO_STREAM) {
        const int qi = ((en.pos0 + t * 64) >> 8) - (en.pos0 >> 8);
        wq = ((const NSK_SCALAR uint32_t *)g.seg_wide)[en.wide_off + (size_t)qi * NSK_WIDE_STRIDE(NCH)] != 0xFFFFFFFFu;
    }
    const uint2 rr = wq ? inf_words_wide(k0, k1, (uint32_t)p, s0, s1)
                        : ((en.zmask_ev >> 16) & 1u) ? inf_words_quad(k0, k1, (uint32_t)p, s0, s1)
                                                     : inf_words(k0, k1, (uint32_t)p, s0, s1);
    const double z0 = nsk_exp(p0);
    const double z1 = z0 + nsk_exp(p1);
    const double z = u53(rr.x, rr.y) * z1;
    const int nv = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    if (v >= 0) {
        g.val[p] = (VT)nv;
        if (!burnin) g.cnt_pos[p] = (uint8_t)(tally + nv);
    }
}

}  // namespace nsk
