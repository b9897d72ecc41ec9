// nsk_api.hip -- sweep kernels (gfx950) and the C-ABI entry points of include/numbskull_amd.h.
//
// Replaces the callee side of the reference's three run_pool(...) call sites
// (numbskull/factorgraph.py:141,163,202): gibbsthread (inference.py:10-33) and
// learnthread/sample_and_sgd (learning.py:12-125).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time with dlopen

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"
#include "nsk_compile.h"
#include "nsk_device.h"

using namespace nsk;

// =============================================================================================
// kernels
// =============================================================================================
#define NSK_BLOCK 256
// persistent grids of the learning kernels (rows of the SMALLW partial-sum tables)
#define NSK_LEARN_FAST_BLOCKS 2048
#define NSK_LEARN_LIST_BLOCKS 512
#define NSK_LEARN_GEN_BLOCKS 2048
#define NSK_LEARN_ROWS (NSK_LEARN_FAST_BLOCKS + NSK_LEARN_LIST_BLOCKS + NSK_LEARN_GEN_BLOCKS)

// One colour class of one inference sweep: lane <-> variable at position pbegin + global lane id.
// gibbsthread's loop body (inference.py:20-33) for that variable.
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_phase(DevGraph<VT> g, int pbegin, int pend,
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
    const u32x4 r = philox4x32(k0, k1, (uint32_t)v, 0u, s0, s1);
    const int nv = draw_sample(g, v, info, g.p_slot[p], g.val, u53(r.x, r.y));
    g.val[v] = (VT)nv;
    if (!burnin) {                                      // inference.py:29-33
        const int base = g.p_cnt[p];
        if (NSK_INFO_CARD(info) == 2) g.cnt[base] += nv;
        else g.cnt[base + nv] += 1;
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
__global__ __launch_bounds__(NSK_BLOCK) void k_refresh_prog_weights(const uint32_t *prog, const double *w,
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

// Shape tile: every lane has its own factor functions and weights (per-lane header words in the
// stream) but all 64 lanes share the word layout -- which words are headers, which are members,
// where entries start and end -- so the walk is driven by a scalar role program and only the
// per-lane facts (function code, weight, member values) are vector work.  This is the shape of
// graphs whose factors carry individual weights.
template <typename VT>
__device__ __forceinline__ void tile_potentials_shape(const DevGraph<VT> &g, const VT *val,
                                                      const uint4 *sp, int len, uint32_t prog,
                                                      double &p0, double &p1) {
    const NSK_SCALAR uint32_t *rp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    uint32_t code = 0;                   // per lane: 0 NOOP 1 IMPLY_NATURAL 2 OR 3 AND/ISTRUE 4 EQUAL
    double w = 0.0;
    int first = 0;
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
        for (int i = 0; i < 4; i++) role[i] = rp[4 * c + i] & 0xFu;      // scalar; 0 = padding word
        double wv[4];
        int xv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {                                    // one gather per word
            wv[i] = 0.0; xv[i] = 0;
            if (role[i] & 1u) wv[i] = g.w[wd[i] & 0xFFFFFFu];
            else if (role[i]) xv[i] = (int)val[wd[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (role[i] & 1u) {                                          // header: open an entry
                code = (0x343210u >> (4u * (wd[i] >> 27))) & 0xFu;       // function+1 in 0..5 -> code
                w = wv[i];
                first = 0; allnz = true; any1 = false; alleq = true;
                if (role[i] & 8u) finish(true);
            } else if (role[i]) {                                        // member
                const int x = xv[i];
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

template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_fast(DevGraph<VT> g, int pbegin, int pend,
                                                          int wb_base, int nblocks,
                                                          const uint32_t *tile_list, int nlist,
                                                          int sample_evidence, int burnin,
                                                          uint32_t k0, uint32_t k1, uint32_t s0,
                                                          uint32_t s1) {
    const int lb = xcd_logical_block((int)blockIdx.x, nblocks);
    if (lb < 0) return;
    const int lane = (int)(threadIdx.x & 63);
    int wave = __builtin_amdgcn_readfirstlane(lb * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (tile_list) {                                      // list mode: the tiles outside segments
        if (wave >= nlist) return;
        wave = (int)__builtin_amdgcn_readfirstlane(tile_list[wave]);
    }
    const int p = pbegin + wave * 64 + lane;
    if (pbegin + wave * 64 >= pend) return;               // whole wave beyond the range
    const int v0 = p < pend ? g.p_vid[p] : -1;            // -1 also marks padding positions
    const bool valid = v0 >= 0;
    const int v = valid ? v0 : 0;
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
        if (kind == 7u) tile_potentials_shape(g, g.val, sp, (int)(td.w & 0xFFu), td.z, p0, p1);
        else if (kind == 4u) tile_potentials_uniform<VT, 4>(g, g.val, sp, len, td.z, p0, p1);
        else if (kind == 0u) tile_potentials_uniform<VT, 0>(g, g.val, sp, len, td.z, p0, p1);
        else if (kind == 2u) tile_potentials_uniform<VT, 2>(g, g.val, sp, len, td.z, p0, p1);
        else tile_potentials_uniform<VT, 3>(g, g.val, sp, len, td.z, p0, p1);
    }
    if (!valid) return;
    const int ev = NSK_INFO_EV(info);
    if (!(ev == 0 || sample_evidence)) return;
    const u32x4 rr = philox4x32(k0, k1, (uint32_t)v, 0u, s0, s1);
    const double z0 = nsk_exp(p0);
    const double z1 = z0 + nsk_exp(p1);
    const double z = u53(rr.x, rr.y) * z1;
    const int nv = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    g.val[v] = (VT)nv;
    if (!burnin) g.cnt_pos[p] = (uint8_t)(tally + nv);
}

// Homogeneous segments: runs of consecutive uniform tiles with one program, slot count, kind and
// evidence flag (the shape-class layout of nsk_compile.cpp makes whole classes such runs).  Up to
// NSK_SEG_MAX segments of one (kind, chunk count) share a launch; everything the descriptor-driven
// kernel fetches per tile comes from the kernel-argument table here, and the body is straight
// line: ids, 16-byte member loads, byte gathers, compares, draw, store.
#define NSK_SEG_MAX 8
struct SegTable {
    int n;
    int tile_start[NSK_SEG_MAX + 1];      // first tile of each segment in this launch's numbering
    int pos0[NSK_SEG_MAX];                // position of the segment's first lane
    uint32_t adj_off[NSK_SEG_MAX];        // stream offset (16-byte units) of its first tile
    uint32_t prog[NSK_SEG_MAX];           // slot program
};

template <typename VT, int KIND, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_gibbs_seg(DevGraph<VT> g, SegTable tab, int nblocks,
                                                         int burnin, uint32_t k0, uint32_t k1,
                                                         uint32_t s0, uint32_t s1) {
    const int lb = xcd_logical_block((int)blockIdx.x, nblocks);
    if (lb < 0) return;
    const int lane = (int)(threadIdx.x & 63);
    const int T = __builtin_amdgcn_readfirstlane(lb * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (T >= tab.tile_start[tab.n]) return;
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < NSK_SEG_MAX; i++) sidx += (i < tab.n && T >= tab.tile_start[i]) ? 1 : 0;
    const int t = T - tab.tile_start[sidx];
    const uint32_t prog = tab.prog[sidx];
    const int p = tab.pos0[sidx] + t * 64 + lane;
    const int v = g.p_vid[p];                             // -1: padding lane at a class end
    const uint8_t tally = burnin ? (uint8_t)0 : g.cnt_pos[p];
    const uint4 *sp = g.adj + tab.adj_off[sidx] + (size_t)t * (64 * NCH) + lane;
    const NSK_SCALAR double *tw = (const NSK_SCALAR double *)(g.prog_w + 2 * (size_t)prog);
    const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    uint4 q[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) q[c] = sp[c * 64];
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
    const u32x4 rr = philox4x32(k0, k1, (uint32_t)v, 0u, s0, s1);
    const double z0 = nsk_exp(p0);
    const double z1 = z0 + nsk_exp(p1);
    const double z = u53(rr.x, rr.y) * z1;
    const int nv = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    if (v >= 0) {
        g.val[v] = (VT)nv;
        if (!burnin) g.cnt_pos[p] = (uint8_t)(tally + nv);
    }
}

// ---------------------------------------------------------------------------------------------
// Learning.  One colour class of one learning sweep = sample_and_sgd (learning.py:46-125) for every
// variable of the class with the weights frozen; gradients go to per-weight fixed-point sums and
// the weight update for the whole class is applied afterwards (DESIGN.md "device-mode learning").
// ---------------------------------------------------------------------------------------------
struct LearnParams {
    int regularization, learn_non_evidence;
    double inv_trunc;
    uint32_t k0, k1, s0, s1;
    int row_base;               // first row of this launch in the partial-sum tables (SMALLW)
};

// per-block accumulation tables in LDS (SMALLW) or the global accumulators
template <bool SMALLW, typename VT>
__device__ __forceinline__ GradSink open_sink(const DevGraph<VT> &g, char *smem) {
    GradSink sk;
    if (SMALLW) {
        const int nw = g.nweight;
        sk.G = (long long *)smem;
        sk.K = (uint32_t *)(smem + 8 * (size_t)nw);
        sk.T = sk.K + nw;
        for (int i = (int)threadIdx.x; i < nw; i += NSK_BLOCK) { sk.G[i] = 0; sk.K[i] = 0; sk.T[i] = 0; }
        __syncthreads();
    } else {
        sk.G = g.G; sk.K = g.K; sk.T = g.T;
    }
    return sk;
}

template <bool SMALLW, typename VT>
__device__ __forceinline__ void close_sink(const DevGraph<VT> &g, const GradSink &sk, int row) {
    if (!SMALLW) return;
    __syncthreads();
    const int nw = g.nweight;
    for (int i = (int)threadIdx.x; i < nw; i += NSK_BLOCK) {
        g.part_G[(size_t)row * nw + i] = sk.G[i];
        g.part_K[(size_t)row * nw + i] = sk.K[i];
        g.part_T[(size_t)row * nw + i] = sk.T[i];
    }
}

// Generic learning kernel.  Work items are 64-position groups: item i covers positions
// pbegin + 64 i (range mode) or list[i] (list mode: the per-lane-header tiles of the fast range),
// clipped at pend.  A persistent grid strides over the items so that SMALLW blocks flush once.
template <typename VT, bool SMALLW>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_phase(DevGraph<VT> g, int pbegin, int pend,
                                                           const uint32_t *list, int nitems,
                                                           LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int wave0 = (int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
    const int nwaves = (int)(gridDim.x * (NSK_BLOCK / 64));
    for (int item = wave0; item < nitems; item += nwaves) {
        const int p = (list ? (int)list[item] : pbegin + 64 * item) + lane;
        bool more = false, truncate = false;
        int v = 0, evidence = 0, proposal = 0, a = 0, ae = 0, b = 0, be = 0;
        if (p < pend && g.p_vid[p] >= 0) {
            const uint32_t info = g.p_info[p];
            const int ev = NSK_INFO_EV(info);
            const int slot0 = g.p_slot[p];
            v = g.p_vid[p];
            const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)v, 0u, lp.s0, lp.s1);
            if (ev != 1) evidence = draw_sample(g, v, info, slot0, g.val_evid, u53(r.z, r.w));   // 54-58
            else evidence = (int)g.p_init[p];                                                     // 61-62
            g.val_evid[v] = (VT)evidence;
            proposal = draw_sample(g, v, info, slot0, g.val, u53(r.x, r.y));                      // 66-70
            g.val[v] = (VT)proposal;
            if (lp.learn_non_evidence || ev == 1) {                                               // 71-72
                if (lp.regularization == 1) {                                                     // 90
                    const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)v, 1u, lp.s0, lp.s1);
                    truncate = u53(t.x, t.y) < lp.inv_trunc;
                }
                const int step = NSK_INFO_DT1(info);
                a = g.slot_off[slot0 + step * evidence];
                ae = g.slot_off[slot0 + step * evidence + 1];
                if (step && evidence != proposal) {
                    b = g.slot_off[slot0 + proposal];
                    be = g.slot_off[slot0 + proposal + 1];
                }
                more = (a < ae) || (b < be);
            }
        }
        // union of the two sorted-unique lists (learning.py:76-95), one factor per iteration; the
        // loop is wave-uniform so that accumulate_gradient can reduce across the wave
        while (__ballot(more)) {
            bool have = false;
            int wid = 0;
            long long gfix = 0;
            if (more) {
                const int fa = a < ae ? g.fidx[a] : 0x7fffffff;
                const int fb = b < be ? g.fidx[b] : 0x7fffffff;
                const int fid = fa < fb ? fa : fb;
                if (fa == fid) a++;
                if (fb == fid) b++;
                more = (a < ae) || (b < be);
                const uint4 rec = g.f_rec[fid];
                wid = (int)rec.z;
                if (!g.w_fixed[wid]) {                                                            // 100-101
                    const double p0 = eval_factor(g, rec, v, evidence, g.val_evid);
                    const double p1 = eval_factor(g, rec, v, proposal, g.val);
                    const double gradient = (p1 - p0) * g.f_feat[fid];                            // 109
                    gfix = __double2ll_rn(gradient * NSK_GRAD_SCALE);
                    have = true;
                }
            }
            accumulate_gradient(sk, have, wid, gfix, truncate);
        }
    }
    close_sink<SMALLW>(g, sk, lp.row_base + (int)blockIdx.x);
}

// "satisfied" bits of one slot for the sampled variable at 0 / at 1; the generic flavour keeps
// the running facts about the entry's other members in `st` (same algebra as slot_step).
__device__ __forceinline__ void slot_sat(SlotState &st, uint32_t s, int x, bool &b0, bool &b1) {
    const bool F = (s >> 27) & 1u, ig = (s >> 29) & 1u;
    const uint32_t code = (s >> 24) & 7u;
    const bool nz = ig || (x != 0), one = !ig && (x == 1);
    st.alleq = F || (st.alleq && (x == st.first));
    st.allnz = (F || st.allnz) && nz;
    st.any1 = (!F && st.any1) || one;
    st.first = F ? x : st.first;
    const bool isEq = code == 4u, isAnd = code == 3u || code == 1u, isOr = code == 2u;
    b0 = (isEq && st.alleq && (ig || st.first == 0)) || (isOr && st.any1);
    b1 = (isEq && st.alleq && (ig || st.first == 1)) || (isAnd && st.allnz) || isOr;
}

template <int CODE>
__device__ __forceinline__ void pair_sat(int x, bool &b0, bool &b1) {
    if (CODE == 4) { b0 = x == 0; b1 = x == 1; }
    else if (CODE == 2) { b0 = x == 1; b1 = true; }
    else { b0 = false; b1 = x != 0; }
}

// One uniform tile of the learning sweep.  Both chains are walked together: x from var_value (free
// chain), xe from var_value_evid; per closing slot the satisfied bits for candidates 0/1 are kept
// in lane bitfields so that, once evidence and proposal are known, the entry's gradient over the
// wave is (hi - lo) * (popcount(free satisfied) - popcount(evidence satisfied)): two scalar
// popcounts, one accumulator update per entry per wave.
template <typename VT, int KIND>
__device__ __forceinline__ void learn_tile(const DevGraph<VT> &g, const GradSink &sk, const uint4 *sp,
                                           int len, uint32_t prog, int p, bool valid,
                                           const LearnParams &lp) {
    const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    const NSK_SCALAR double *tw = (const NSK_SCALAR double *)(g.prog_w + 2 * (size_t)prog);
    const uint32_t info = valid ? g.p_info[p] : 0u;
    const int v = valid ? g.p_vid[p] : 0;
    const int ev = NSK_INFO_EV(info);
    const int init = valid ? (int)g.p_init[p] : 0;
    const bool need_evid = __ballot(valid && ev != 1) != 0;          // wave-uniform

    double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
    uint32_t B0 = 0, B1 = 0, C0 = 0, C1 = 0;                           // bit i: slot i satisfied
    SlotState sf = {0, true, false, true}, se = {0, true, false, true};
    uint32_t sl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) sl[i] = pp[i];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half * 4 < len) {                                            // scalar
            const uint4 q = sp[half * 64];
            const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
            int x[4], xe[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] = (int)g.val[wd[i]]; xe[i] = (int)g.val_evid[wd[i]]; }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = half * 4 + i;
                const double thi = tw[2 * j], tlo = tw[2 * j + 1];
                bool b0, b1, c0, c1;
                if (KIND) { pair_sat<KIND>(x[i], b0, b1); pair_sat<KIND>(xe[i], c0, c1); }
                else { slot_sat(sf, sl[j], x[i], b0, b1); slot_sat(se, sl[j], xe[i], c0, c1); }
                p0 = p0 + (b0 ? thi : tlo);
                p1 = p1 + (b1 ? thi : tlo);
                if (need_evid) {
                    q0 = q0 + (c0 ? thi : tlo);
                    q1 = q1 + (c1 ? thi : tlo);
                }
                B0 |= (b0 ? 1u : 0u) << j; B1 |= (b1 ? 1u : 0u) << j;
                C0 |= (c0 ? 1u : 0u) << j; C1 |= (c1 ? 1u : 0u) << j;
            }
        }
    }
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)v, 0u, lp.s0, lp.s1);
    int evidence = init;                                                  // learning.py:61-62
    if (need_evid && ev != 1) {                                           // 54-58
        const double z0 = nsk_exp(q0), z1 = z0 + nsk_exp(q1);
        const double z = u53(r.z, r.w) * z1;
        evidence = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    }
    const double z0 = nsk_exp(p0), z1 = z0 + nsk_exp(p1);                 // 66-70
    const double z = u53(r.x, r.y) * z1;
    const int proposal = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    if (valid) {
        g.val_evid[v] = (VT)evidence;
        g.val[v] = (VT)proposal;
    }
    const bool part = valid && (lp.learn_non_evidence || ev == 1);        // 71-72
    bool truncate = false;
    if (lp.regularization == 1) {                                         // 90
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)v, 1u, lp.s0, lp.s1);
        truncate = part && (u53(t.x, t.y) < lp.inv_trunc);
    }
    const unsigned long long pm = __ballot(part);
    if (pm == 0) return;
    const uint32_t satf = proposal ? B1 : B0, sate = evidence ? C1 : C0;
    const uint32_t nk = (uint32_t)__popcll(pm), nt = (uint32_t)__popcll(__ballot(truncate));
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t s = sl[j];
        const bool closes = (s >> 28) & 1u, fixed = (s >> 30) & 1u;      // uniform
        if (j < len && closes && !fixed) {
            const uint32_t code = (s >> 24) & 7u;
            const long long span = code == 0u ? 0 : (code == 1u ? 1 : 2);            // hi - lo
            const int nf = __popcll(__ballot(part && ((satf >> j) & 1u)));
            const int ne = __popcll(__ballot(part && ((sate >> j) & 1u)));
            if ((threadIdx.x & 63) == 0) {
                const long long dG = (span * (long long)(nf - ne)) << 32;            // Q31.32
                const int wid = (int)(s & 0xFFFFFFu);
                atomicAdd((unsigned long long *)&sk.G[wid], (unsigned long long)dG);
                atomicAdd(&sk.K[wid], nk);
                if (nt) atomicAdd(&sk.T[wid], nt);
            }
        }
    }
}

// Learning over the uniform tiles of a colour class (tiles with per-lane headers are left to
// k_learn_phase in list mode).  Each wave takes a contiguous run of tiles.
template <typename VT, bool SMALLW>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_fast(DevGraph<VT> g, int pbegin, int pend,
                                                          int wb_base, int ntiles, LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6)));
    const int nwaves = (int)(gridDim.x * (NSK_BLOCK / 64));
    const int per = (ntiles + nwaves - 1) / nwaves;
    const int t1 = min(ntiles, (wave0 + 1) * per);
    for (int t = wave0 * per; t < t1; t++) {
        const NSK_SCALAR uint32_t *tdp = (const NSK_SCALAR uint32_t *)(g.tiles + (wb_base + t));
        const struct { uint32_t x, y, z, w; } td = {tdp[0], tdp[1], tdp[2], tdp[3]};
        if (td.z == NSK_PAD_WORD || ((td.w >> 8) & 7u) == 7u) continue;   // per-lane headers: generic kernel's job
        const int p = pbegin + t * 64 + lane;
        const bool valid = p < pend && g.p_vid[p] >= 0;
        const uint4 *sp = g.adj + td.x + lane;
        const uint32_t kind = (td.w >> 8) & 7u;
        if (kind == 4u) learn_tile<VT, 4>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
        else if (kind == 0u) learn_tile<VT, 0>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
        else if (kind == 2u) learn_tile<VT, 2>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
        else learn_tile<VT, 3>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
    }
    close_sink<SMALLW>(g, sk, lp.row_base + (int)blockIdx.x);
}

// The weight update of learning.py:110-125 applied to a whole colour class at once
// (DESIGN.md "device-mode learning" gives the closed forms; the oracle restates them).
__device__ __forceinline__ double apply_update(double x, long long G, uint32_t k, uint32_t t, double step,
                                               int regularization, double reg_param, double truncation) {
    const double Gf = (double)G * (1.0 / 4294967296.0);
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

__global__ __launch_bounds__(NSK_BLOCK) void k_apply_weights(double *w, long long *G, uint32_t *K,
                                                             uint32_t *T, int nweight, double step,
                                                             int regularization, double reg_param,
                                                             double truncation) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= nweight) return;
    const uint32_t k = K[i];
    if (k == 0) return;
    w[i] = apply_update(w[i], G[i], k, T[i], step, regularization, reg_param, truncation);
    G[i] = 0; K[i] = 0; T[i] = 0;
}

// SMALLW flavour: one block per weight adds up the per-block rows, applies the update and rewrites
// the prog_w entries of the slot programs that use this weight (so no separate refresh launch).
__global__ __launch_bounds__(NSK_BLOCK) void k_apply_weights_rows(double *w, const long long *part_G,
                                                                  const uint32_t *part_K, const uint32_t *part_T,
                                                                  int nrows, int nweight, double step,
                                                                  int regularization, double reg_param,
                                                                  double truncation, const uint32_t *prog,
                                                                  double *prog_w, int nprog) {
    __shared__ long long red[3][NSK_BLOCK / 64];
    __shared__ double wnew;
    const int i = (int)blockIdx.x, tid = (int)threadIdx.x;
    long long G = 0, K = 0, T = 0;
    for (int r = tid; r < nrows; r += NSK_BLOCK) {
        G += part_G[(size_t)r * nweight + i];
        K += (long long)part_K[(size_t)r * nweight + i];
        T += (long long)part_T[(size_t)r * nweight + i];
    }
    G = wave_sum_i64(G); K = wave_sum_i64(K); T = wave_sum_i64(T);
    if ((tid & 63) == 0) { red[0][tid >> 6] = G; red[1][tid >> 6] = K; red[2][tid >> 6] = T; }
    __syncthreads();
    if (tid == 0) {
        G = 0; K = 0; T = 0;
        for (int k = 0; k < NSK_BLOCK / 64; k++) { G += red[0][k]; K += red[1][k]; T += red[2][k]; }
        double x = w[i];
        if (K > 0) {
            x = apply_update(x, G, (uint32_t)K, (uint32_t)T, step, regularization, reg_param, truncation);
            w[i] = x;
        }
        wnew = x;
    }
    __syncthreads();
    const double x = wnew;
    for (int j = tid; j < nprog; j += NSK_BLOCK) {
        const uint32_t s = prog[j];
        if ((s >> 31) || (int)(s & 0xFFFFFFu) != i) continue;
        const uint32_t code = (s >> 24) & 7u;
        const bool last = (s >> 28) & 1u;
        const double hi = code == 0u ? 0.0 : 1.0;
        const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
        prog_w[2 * j] = last ? x * hi : 0.0;
        prog_w[2 * j + 1] = last ? x * lo : 0.0;
    }
}

// int32 per-call tally deltas -> int64 master copy (the host-visible `count`)
__global__ __launch_bounds__(NSK_BLOCK) void k_fold_counts(int32_t *delta, long long *total, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= n) return;
    total[i] += (long long)delta[i];
    delta[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(NSK_BLOCK) void k_stream_copy(const T *__restrict__ src, T *__restrict__ dst,
                                                           long long n) {
    const long long stride = (long long)gridDim.x * NSK_BLOCK;
    for (long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// ---- boundary exchange ---------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_exchange_pack(const VT *val, const int32_t *send_vids,
                                                             VT *sendbuf, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) sendbuf[i] = val[send_vids[i]];
}

// recv_src[j] = rank that owns recv_vids[j]; entries of this rank itself are skipped
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_exchange_unpack(VT *val, const int32_t *recv_vids,
                                                               const int32_t *recv_slot, const VT *recvbuf,
                                                               int n) {
    const int j = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (j >= n) return;
    const int sl = recv_slot[j];            // index into the gathered buffer, -1 = own entry
    if (sl >= 0) val[recv_vids[j]] = recvbuf[sl];
}

// weight merge of the partitioned learning sweep: delta = w - start ... w = start + sum(delta)
__global__ __launch_bounds__(NSK_BLOCK) void k_weight_delta(const double *w, const double *start, double *delta, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) delta[i] = w[i] - start[i];
}
__global__ __launch_bounds__(NSK_BLOCK) void k_weight_merge(double *w, const double *start, const double *delta, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) w[i] = start[i] + delta[i];
}

__global__ void k_selftest_exp(const double *x, double *y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = nsk_exp(x[i]);
}

__global__ void k_selftest_philox(uint32_t k0, uint32_t k1, uint32_t stream, uint32_t s0, uint32_t s1,
                                  long long n, uint32_t *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = philox4x32(k0, k1, (uint32_t)i, stream, s0, s1);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// position-indexed tally deltas of the fast path -> int64 master copy at cstart[vid]
__global__ __launch_bounds__(NSK_BLOCK) void k_fold_counts_pos(uint8_t *cnt_pos, const int32_t *p_cnt,
                                                               long long *total, int npos) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= npos) return;
    const int d = cnt_pos[i];
    if (d) {
        total[p_cnt[i]] += (long long)d;
        cnt_pos[i] = 0;
    }
}

// ---- sequential validation scan: one lane walks variable ids in order with MT19937 -----------
template <typename VT>
__global__ void k_seq_gibbs(DevGraph<VT> g, const int32_t *v_pos, MTState *np_rng, int nsweeps,
                            int sample_evidence, int burnin) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int s = 0; s < nsweeps; s++) {
        for (int v = 0; v < g.nvar; v++) {
            const int p = v_pos[v];
            if (p < 0) continue;
            const uint32_t info = g.p_info[p];
            if (!(NSK_INFO_EV(info) == 0 || sample_evidence)) continue;
            // the reference fills Z first and draws its uniform afterwards; draw_sample only needs
            // u at the very end, and nothing else consumes the stream in between
            const double u = mt_res53(np_rng);
            const int nv = draw_sample(g, v, info, g.p_slot[p], g.val, u);
            g.val[v] = (VT)nv;
            if (!burnin) {
                const int base = g.p_cnt[p];
                if (NSK_INFO_CARD(info) == 2) g.cnt[base] += nv;
                else g.cnt[base + nv] += 1;
            }
        }
    }
}

template <typename VT>
__global__ void k_seq_learn(DevGraph<VT> g, const int32_t *v_pos, MTState *np_rng, MTState *py_rng,
                            int nsweeps, double step, double decay, int regularization,
                            double reg_param, double truncation, int learn_non_evidence) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int s = 0; s < nsweeps; s++) {
        for (int v = 0; v < g.nvar; v++) {
            const int p = v_pos[v];
            if (p < 0) continue;
            const uint32_t info = g.p_info[p];
            const int ev = NSK_INFO_EV(info);
            const int slot0 = g.p_slot[p];
            int evidence;
            if (ev != 1) evidence = draw_sample(g, v, info, slot0, g.val_evid, mt_res53(np_rng));
            else evidence = (int)g.p_init[p];
            g.val_evid[v] = (VT)evidence;
            const int proposal = draw_sample(g, v, info, slot0, g.val, mt_res53(np_rng));
            g.val[v] = (VT)proposal;
            if (!learn_non_evidence && ev != 1) continue;
            const int st = NSK_INFO_DT1(info);
            int a = g.slot_off[slot0 + st * evidence], ae = g.slot_off[slot0 + st * evidence + 1];
            int b = 0, be = 0;
            if (st && evidence != proposal) {
                b = g.slot_off[slot0 + proposal];
                be = g.slot_off[slot0 + proposal + 1];
            }
            bool truncate = false;
            if (regularization == 1) truncate = mt_res53(py_rng) < 1.0 / truncation;
            while (a < ae || b < be) {          // sorted, de-duplicated union == learning.py:76-98
                const int fa = a < ae ? g.fidx[a] : 0x7fffffff;
                const int fb = b < be ? g.fidx[b] : 0x7fffffff;
                const int fid = fa < fb ? fa : fb;
                if (fa == fid) a++;
                if (fb == fid) b++;
                const uint4 rec = g.f_rec[fid];
                const int wid = (int)rec.z;
                if (g.w_fixed[wid]) continue;
                const double p0 = eval_factor(g, rec, v, evidence, g.val_evid);
                const double p1 = eval_factor(g, rec, v, proposal, g.val);
                const double gradient = (p1 - p0) * g.f_feat[fid];
                double w = g.w[wid];
                if (regularization == 2) {
                    w *= (1.0 / (1.0 + reg_param * step));
                    w -= step * gradient;
                } else if (regularization == 1) {
                    w -= step * gradient;
                    if (truncate) {
                        const double l1delta = reg_param * step * truncation;
                        w = (w > 0) ? fmax(0.0, w - l1delta) : fmin(0.0, w + l1delta);
                    }
                } else {
                    w -= step * gradient;
                }
                g.w[wid] = w;
            }
        }
        step *= decay;
    }
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
namespace nsk { void set_error(const std::string &m) { g_err = m; } }

// RCCL entry points, bound at run time (nsk_comm_init)
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;



#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(NSK_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

struct nsk_graph {
    Compiled c;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<void *> allocs;
    int64_t device_bytes = 0;
    // device arrays
    int32_t *p_vid = nullptr, *p_slot = nullptr, *p_cnt = nullptr, *slot_off = nullptr, *fidx = nullptr;
    uint32_t *p_info = nullptr, *f_rec = nullptr;
    void *p_init = nullptr;
    int32_t *m_rec = nullptr, *v_card = nullptr,
            *v_pos = nullptr;
    double *f_feat = nullptr, *w = nullptr, *logtab = nullptr;
    uint8_t *w_fixed = nullptr;
    void *val = nullptr, *val_evid = nullptr;
    int32_t *cnt = nullptr;
    uint8_t *cnt_pos = nullptr;
    int pos_tally_sweeps = 0;      // sweeps accumulated in the uint8 position tally
    uint32_t *adj = nullptr, *tiles = nullptr, *tile_hdr = nullptr;
    double *prog_w = nullptr;
    uint32_t *dyn_tiles = nullptr, *rest_tiles = nullptr;
    long long *part_G = nullptr;       // SMALLW: rows of per-block partial sums
    uint32_t *part_K = nullptr, *part_T = nullptr;
    bool smallw = false;
    bool weights_dirty = true;      // prog_w must be rebuilt before the next fast-path launch
    bool weights_exposed = false;   // the weight buffer was handed out: assume it changes between calls
    // boundary exchange (multi-GPU)
    int xworld = 0, xrank = 0;
    int64_t xslot = 0, xnsend = 0, xnrecv = 0;
    int32_t *x_send_vids = nullptr, *x_recv_vids = nullptr, *x_recv_slot = nullptr;
    void *x_send = nullptr, *x_recv = nullptr, *x_send_evid = nullptr, *x_recv_evid = nullptr;
    double *w_start = nullptr, *w_delta = nullptr;
    // native RCCL
    void *rccl_lib = nullptr, *rccl_comm = nullptr;
    long long *cnt_total = nullptr, *G = nullptr;
    uint32_t *K = nullptr, *T = nullptr;
    MTState *mt_np = nullptr, *mt_py = nullptr;
    // run state
    uint64_t seed = 0, sweep = 0;
    int scan = NSK_SCAN_CHROMATIC;
    bool cnt_dirty = false;
    int64_t sweeps_done = 0;
    // profiling bracket
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t launches = 0, launches_at_begin = 0;
};

template <typename T>
static int dev_alloc(nsk_graph *g, T **ptr, size_t n) {
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(NSK_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    g->allocs.push_back(p);
    g->device_bytes += (int64_t)bytes;
    *ptr = (T *)p;
    return NSK_OK;
}

template <typename T>
static int dev_upload(nsk_graph *g, T **ptr, const std::vector<T> &h) {
    int rc = dev_alloc(g, ptr, h.size());
    if (rc) return rc;
    if (!h.empty()) HIPCHECK(hipMemcpyAsync(*ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, g->stream));
    return NSK_OK;
}

// narrow int32 host values to the device value type (int8 / int32) and upload
static int upload_values(nsk_graph *g, void *dst, const int32_t *src, size_t n) {
    if (g->c.vbytes == 4) {
        if (n) HIPCHECK(hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, g->stream));
        HIPCHECK(hipStreamSynchronize(g->stream));
        return NSK_OK;
    }
    std::vector<int8_t> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = (int8_t)src[i];
    if (n) HIPCHECK(hipMemcpyAsync(dst, tmp.data(), n, hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

static void mt_seed_numpy(MTState &s, uint32_t seed) {      // np.random.seed(int): init_genrand
    s.mt[0] = seed;
    for (int i = 1; i < 624; i++) s.mt[i] = 1812433253u * (s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) + (uint32_t)i;
    s.idx = 624;
}

static void mt_seed_python(MTState &s, uint64_t seed) {     // random.seed(int): init_by_array
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int keylen = key[1] ? 2 : 1;
    mt_seed_numpy(s, 19650218u);
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        s.mt[i] = (s.mt[i] ^ ((s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s.mt[0] = s.mt[623]; i = 1; }
        if (j >= keylen) j = 0;
    }
    for (int k = 623; k; k--) {
        s.mt[i] = (s.mt[i] ^ ((s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s.mt[0] = s.mt[623]; i = 1; }
    }
    s.mt[0] = 0x80000000u;
    s.idx = 624;
}

template <typename VT>
static DevGraph<VT> view(nsk_graph *g) {
    DevGraph<VT> d;
    d.p_vid = g->p_vid; d.p_info = g->p_info; d.p_slot = g->p_slot; d.p_cnt = g->p_cnt;
    d.p_init = (const VT *)g->p_init;
    d.slot_off = g->slot_off; d.fidx = g->fidx;
    d.f_rec = (const uint4 *)g->f_rec; d.f_feat = g->f_feat;
    d.m_rec = (const int2 *)g->m_rec; d.v_card = g->v_card;
    d.w = g->w; d.w_fixed = g->w_fixed; d.logtab = g->logtab;
    d.val = (VT *)g->val; d.val_evid = (VT *)g->val_evid; d.cnt = g->cnt;
    d.G = g->G; d.K = g->K; d.T = g->T;
    d.adj = (const uint4 *)g->adj; d.tiles = (const uint4 *)g->tiles; d.tile_hdr = g->tile_hdr;
    d.prog_w = g->prog_w;
    d.part_G = g->part_G; d.part_K = g->part_K; d.part_T = g->part_T;
    d.nweight = (int32_t)g->c.nweight;
    d.cnt_pos = g->cnt_pos;
    d.nvar = (int32_t)g->c.nvar;
    d.head_by_vid = (g->c.flags & NSK_FLAG_HEAD_BY_VID) ? 1 : 0;
    return d;
}

extern "C" {

const char *nsk_last_error(void) { return g_err.c_str(); }
const char *nsk_version(void) { return "numbskull_amd 0.1.1 (gfx950)"; }

int nsk_device_count(int *count) {
    if (!count) return fail(NSK_E_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(NSK_E_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return NSK_OK;
}

int nsk_graph_destroy(nsk_graph *g) {
    if (!g) return NSK_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    for (void *p : g->allocs) (void)hipFree(p);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)g->rccl_comm);
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return NSK_OK;
}

static int create_impl(const nsk_graph_desc *desc, nsk_graph *g) {
    std::string err;
    int rc = compile_graph(desc, g->c, err);
    if (rc) return fail(rc, err);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NSK_E_DEVICE, "no HIP device available: the Gibbs sweep runs on the GPU only "
                                  "(there is no CPU fallback)");
    if (desc->device < 0 || desc->device >= ndev) return fail(NSK_E_INVALID, "device ordinal out of range");
    g->device = desc->device;
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->own_stream = true;
    HIPCHECK(hipEventCreate(&g->ev0));
    HIPCHECK(hipEventCreate(&g->ev1));
    Compiled &c = g->c;
#define UP(name) do { rc = dev_upload(g, &g->name, c.name); if (rc) return rc; } while (0)
    UP(p_vid); UP(p_info); UP(p_slot); UP(p_cnt); UP(slot_off); UP(fidx);
    UP(f_rec); UP(f_feat); UP(m_rec); UP(v_card); UP(v_pos);
    UP(w_fixed); UP(logtab); UP(adj); UP(tiles); UP(tile_hdr); UP(dyn_tiles); UP(rest_tiles);
#undef UP
    rc = dev_upload(g, &g->w, c.w_init); if (rc) return rc;
    const size_t nvar = (size_t)c.nvar, npos = (size_t)c.npos, vb = (size_t)c.vbytes;
    uint8_t *tmp = nullptr;
    rc = dev_alloc(g, &tmp, npos * vb); if (rc) return rc; g->p_init = tmp;
    rc = dev_alloc(g, &tmp, nvar * vb); if (rc) return rc; g->val = tmp;
    rc = dev_alloc(g, &tmp, nvar * vb); if (rc) return rc; g->val_evid = tmp;
    rc = upload_values(g, g->p_init, c.p_init.data(), npos); if (rc) return rc;
    rc = upload_values(g, g->val, c.v_init.data(), nvar); if (rc) return rc;
    rc = upload_values(g, g->val_evid, c.v_init.data(), nvar); if (rc) return rc;
    rc = dev_alloc(g, &g->cnt, (size_t)c.ncount); if (rc) return rc;
    rc = dev_alloc(g, &g->cnt_total, (size_t)c.ncount); if (rc) return rc;
    rc = dev_alloc(g, &g->cnt_pos, (size_t)c.npos); if (rc) return rc;
    rc = dev_alloc(g, &g->prog_w, 2 * c.tile_hdr.size()); if (rc) return rc;
    g->smallw = c.nweight > 0 && c.nweight <= NSK_SMALLW;
    if (g->smallw) {
        const size_t cells = (size_t)NSK_LEARN_ROWS * (size_t)c.nweight;
        rc = dev_alloc(g, &g->part_G, cells); if (rc) return rc;
        rc = dev_alloc(g, &g->part_K, cells); if (rc) return rc;
        rc = dev_alloc(g, &g->part_T, cells); if (rc) return rc;
    }
    HIPCHECK(hipMemsetAsync(g->cnt_pos, 0, (c.npos ? c.npos : 1), g->stream));
    rc = dev_alloc(g, &g->G, (size_t)c.nweight); if (rc) return rc;
    rc = dev_alloc(g, &g->K, (size_t)c.nweight); if (rc) return rc;
    rc = dev_alloc(g, &g->T, (size_t)c.nweight); if (rc) return rc;
    rc = dev_alloc(g, &g->mt_np, 1); if (rc) return rc;
    rc = dev_alloc(g, &g->mt_py, 1); if (rc) return rc;
    HIPCHECK(hipMemsetAsync(g->cnt, 0, (c.ncount ? c.ncount : 1) * sizeof(int32_t), g->stream));
    HIPCHECK(hipMemsetAsync(g->cnt_total, 0, (c.ncount ? c.ncount : 1) * sizeof(long long), g->stream));
    HIPCHECK(hipMemsetAsync(g->G, 0, (c.nweight ? c.nweight : 1) * sizeof(long long), g->stream));
    HIPCHECK(hipMemsetAsync(g->K, 0, (c.nweight ? c.nweight : 1) * sizeof(uint32_t), g->stream));
    HIPCHECK(hipMemsetAsync(g->T, 0, (c.nweight ? c.nweight : 1) * sizeof(uint32_t), g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return nsk_set_seed(g, 0, 0);
}

int nsk_graph_create(const nsk_graph_desc *desc, nsk_graph **out) {
    if (!desc || !out) return fail(NSK_E_INVALID, "null argument");
    *out = nullptr;
    nsk_graph *g = new nsk_graph();
    int rc = create_impl(desc, g);
    if (rc) {
        std::string keep = g_err;
        nsk_graph_destroy(g);
        g_err = keep;
        return rc;
    }
    *out = g;
    return NSK_OK;
}

int nsk_set_seed(nsk_graph *g, uint64_t seed, uint64_t sweep0) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    g->seed = seed;
    g->sweep = sweep0;
    MTState a, b;
    mt_seed_numpy(a, (uint32_t)seed);
    mt_seed_python(b, seed);
    HIPCHECK(hipMemcpyAsync(g->mt_np, &a, sizeof(MTState), hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipMemcpyAsync(g->mt_py, &b, sizeof(MTState), hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

int nsk_set_scan(nsk_graph *g, int scan) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (scan != NSK_SCAN_CHROMATIC && scan != NSK_SCAN_SEQUENTIAL) return fail(NSK_E_INVALID, "unknown scan order");
    if (scan == NSK_SCAN_SEQUENTIAL && (g->c.own_begin != 0 || g->c.own_end != g->c.nvar))
        return fail(NSK_E_INVALID, "sequential scan needs the whole graph on one handle");
    g->scan = scan;
    return NSK_OK;
}

int nsk_set_stream(nsk_graph *g, void *hip_stream) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipStreamSynchronize(g->stream));
    if (g->own_stream) { HIPCHECK(hipStreamDestroy(g->stream)); g->own_stream = false; }
    g->stream = (hipStream_t)hip_stream;
    return NSK_OK;
}

int nsk_synchronize(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

// the fast path reads weights through prog_w: rebuild it whenever weights may have changed (start
// of every sweep call -- the host may have written the weight buffer -- and after every update)
static void refresh_prog_weights(nsk_graph *g, bool force = false) {
    if (!force && !g->weights_dirty && !g->weights_exposed) return;
    g->weights_dirty = false;
    const int n = (int)g->c.tile_hdr.size();
    if (n > 0 && g->c.nfast > 0 && g->c.nweight > 0)
        k_refresh_prog_weights<<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->tile_hdr, g->w, g->prog_w, n);
}

static int fold_position_tally(nsk_graph *g) {
    const int np = (int)g->c.npos;
    if (np > 0 && g->c.nfast > 0 && g->pos_tally_sweeps > 0)
        k_fold_counts_pos<<<dim3((np + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->cnt_pos, g->p_cnt, g->cnt_total, np);
    g->pos_tally_sweeps = 0;
    return NSK_OK;
}

static int fold_counts(nsk_graph *g) {
    if (!g->cnt_dirty) return NSK_OK;
    const int n = (int)g->c.ncount;
    if (n > 0)
        k_fold_counts<<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->cnt, g->cnt_total, n);
    fold_position_tally(g);
    HIPCHECK(hipGetLastError());
    g->cnt_dirty = false;
    return NSK_OK;
}

}  // extern "C"

template <typename VT>
static int gibbs_impl(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    DevGraph<VT> d = view<VT>(g);
    if (g->scan == NSK_SCAN_SEQUENTIAL) {
        k_seq_gibbs<VT><<<dim3(1), dim3(64), 0, g->stream>>>(d, g->v_pos, g->mt_np, (int)nsweeps,
                                                            sample_evidence, burnin);
        HIPCHECK(hipGetLastError());
        g->launches++;
        g->sweep += (uint64_t)nsweeps;
    } else {
        const size_t nphase = g->c.phase_start.size() - 1;
        refresh_prog_weights(g);
        for (int64_t s = 0; s < nsweeps; s++) {
            for (size_t ph = 0; ph < nphase; ph++) {
                const int fb = (int)g->c.phase_start[ph], fe = (int)g->c.phase_fast_end[ph];
                const int e = (int)g->c.phase_start[ph + 1];
                if (fe > fb) {      // inlined-adjacency kernels
                    const uint32_t K0 = (uint32_t)g->seed, K1 = (uint32_t)(g->seed >> 32);
                    const uint32_t S0 = (uint32_t)g->sweep, S1 = (uint32_t)(g->sweep >> 32);
                    // segments of this colour, batched by (kind, chunks) into table launches
                    for (int kind = 0; kind <= 4; kind++) {
                        if (kind == 1) continue;                 // IMPLY_NATURAL shares the AND step (3)
                        for (int nch = 1; nch <= 2; nch++) {
                            SegTable tab;
                            tab.n = 0; tab.tile_start[0] = 0;
                            auto flush = [&]() {
                                if (tab.n == 0) return;
                                const int nb = (tab.tile_start[tab.n] + 3) / 4;
                                const dim3 grid(8 * ((nb + 7) / 8)), block(NSK_BLOCK);
#define NSK_SEG(KIND, NCH) k_gibbs_seg<VT, KIND, NCH><<<grid, block, 0, g->stream>>>(d, tab, nb, burnin, K0, K1, S0, S1)
                                if (kind == 4) { if (nch == 1) NSK_SEG(4, 1); else NSK_SEG(4, 2); }
                                else if (kind == 2) { if (nch == 1) NSK_SEG(2, 1); else NSK_SEG(2, 2); }
                                else if (kind == 0) { if (nch == 1) NSK_SEG(0, 1); else NSK_SEG(0, 2); }
                                else { if (nch == 1) NSK_SEG(3, 1); else NSK_SEG(3, 2); }
#undef NSK_SEG
                                g->launches++;
                                tab.n = 0;
                            };
                            for (const Compiled::Segment &sg : g->c.segments) {
                                if (sg.phase != (int)ph) continue;
                                const int k3 = sg.kind == 1 ? 3 : (int)sg.kind;
                                if (k3 != kind || (sg.nslots > 4 ? 2 : 1) != nch) continue;
                                if (!(sg.ev == 0 || sample_evidence)) continue;      // inference.py:24
                                tab.pos0[tab.n] = (int)sg.pos0;
                                tab.adj_off[tab.n] = sg.adj_off;
                                tab.prog[tab.n] = sg.prog;
                                tab.tile_start[tab.n + 1] = tab.tile_start[tab.n] + sg.ntiles;
                                if (++tab.n == NSK_SEG_MAX) flush();
                            }
                            flush();
                        }
                    }
                    const int nrest = (int)(g->c.phase_rest_base[ph + 1] - g->c.phase_rest_base[ph]);
                    if (nrest > 0) {
                        const int nblocks = (nrest + 3) / 4;
                        k_gibbs_fast<VT><<<dim3(8 * ((nblocks + 7) / 8)), dim3(NSK_BLOCK), 0, g->stream>>>(
                            d, fb, fe, (int)g->c.phase_wb_base[ph], nblocks,
                            g->rest_tiles + g->c.phase_rest_base[ph], nrest, sample_evidence, burnin,
                            K0, K1, S0, S1);
                        g->launches++;
                    }
                }
                if (e > fe) {       // generic CSR kernel
                    k_gibbs_phase<VT><<<dim3((e - fe + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                        d, fe, e, sample_evidence, burnin, (uint32_t)g->seed, (uint32_t)(g->seed >> 32),
                        (uint32_t)g->sweep, (uint32_t)(g->sweep >> 32));
                    g->launches++;
                }
            }
            g->sweep++;
            if (!burnin && ++g->pos_tally_sweeps == 255) fold_position_tally(g);   // uint8 tally is full
        }
        HIPCHECK(hipGetLastError());
    }
    if (!burnin) g->cnt_dirty = true;
    g->sweeps_done += nsweeps;
    return NSK_OK;
}

extern "C" int nsk_gibbs_sweeps(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    if (nsweeps == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(g->device));
    return g->c.vbytes == 1 ? gibbs_impl<int8_t>(g, nsweeps, sample_evidence, burnin)
                            : gibbs_impl<int32_t>(g, nsweeps, sample_evidence, burnin);
}

template <typename VT, bool SMALLW>
static int learn_chromatic(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                           double reg_param, int64_t truncation, int learn_non_evidence) {
    DevGraph<VT> d = view<VT>(g);
    const size_t nphase = g->c.phase_start.size() - 1;
    const int nw = (int)g->c.nweight;
    const size_t shmem = SMALLW ? (size_t)nw * 16 : 0;
    LearnParams lp;
    lp.regularization = regularization;
    lp.learn_non_evidence = learn_non_evidence;
    lp.inv_trunc = 1.0 / (double)truncation;
    lp.k0 = (uint32_t)g->seed; lp.k1 = (uint32_t)(g->seed >> 32);
    refresh_prog_weights(g);
    for (int64_t s = 0; s < nsweeps; s++) {
        lp.s0 = (uint32_t)g->sweep; lp.s1 = (uint32_t)(g->sweep >> 32);
        for (size_t ph = 0; ph < nphase; ph++) {
            const int fb = (int)g->c.phase_start[ph], fe = (int)g->c.phase_fast_end[ph];
            const int e = (int)g->c.phase_start[ph + 1];
            if (e <= fb) continue;
            int rows = 0;
            const int ntiles = (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]);
            const int ndyn = (int)(g->c.phase_dyn_base[ph + 1] - g->c.phase_dyn_base[ph]);
            if (ntiles > ndyn) {        // uniform tiles: inlined-adjacency learning kernel
                const int grid = std::min(NSK_LEARN_FAST_BLOCKS, (ntiles + 3) / 4);
                lp.row_base = rows;
                k_learn_fast<VT, SMALLW><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(
                    d, fb, fe, (int)g->c.phase_wb_base[ph], ntiles, lp);
                rows += grid;
                g->launches++;
            }
            if (ndyn > 0) {             // tiles with per-lane headers: generic kernel, list mode
                const int grid = std::min(NSK_LEARN_LIST_BLOCKS, (ndyn + 3) / 4);
                lp.row_base = rows;
                k_learn_phase<VT, SMALLW><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(
                    d, fb, fe, g->dyn_tiles + g->c.phase_dyn_base[ph], ndyn, lp);
                rows += grid;
                g->launches++;
            }
            if (e > fe) {               // variables outside the fast path: generic kernel, range mode
                const int nitems = (e - fe + 63) / 64;
                const int grid = std::min(NSK_LEARN_GEN_BLOCKS, (nitems + 3) / 4);
                lp.row_base = rows;
                k_learn_phase<VT, SMALLW><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(
                    d, fe, e, nullptr, nitems, lp);
                rows += grid;
                g->launches++;
            }
            if (nw > 0) {
                if (SMALLW) {
                    k_apply_weights_rows<<<dim3(nw), dim3(NSK_BLOCK), 0, g->stream>>>(
                        g->w, g->part_G, g->part_K, g->part_T, rows, nw, step, regularization, reg_param,
                        (double)truncation, g->tile_hdr, g->prog_w,
                        g->c.nfast > 0 ? (int)g->c.tile_hdr.size() : 0);
                } else {
                    k_apply_weights<<<dim3((nw + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                        g->w, g->G, g->K, g->T, nw, step, regularization, reg_param, (double)truncation);
                    refresh_prog_weights(g, true);
                }
            }
        }
        g->sweep++;
        step *= decay;                                   // factorgraph.py:206
    }
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

template <typename VT>
static int learn_impl(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                      double reg_param, int64_t truncation, int learn_non_evidence) {
    if (g->scan == NSK_SCAN_SEQUENTIAL) {
        DevGraph<VT> d = view<VT>(g);
        k_seq_learn<VT><<<dim3(1), dim3(64), 0, g->stream>>>(d, g->v_pos, g->mt_np, g->mt_py, (int)nsweeps,
                                                            step, decay, regularization, reg_param,
                                                            (double)truncation, learn_non_evidence);
        HIPCHECK(hipGetLastError());
        g->launches++;
        g->sweep += (uint64_t)nsweeps;
    } else {
        int rc = g->smallw ? learn_chromatic<VT, true>(g, nsweeps, step, decay, regularization, reg_param,
                                                       truncation, learn_non_evidence)
                           : learn_chromatic<VT, false>(g, nsweeps, step, decay, regularization, reg_param,
                                                        truncation, learn_non_evidence);
        if (rc) return rc;
    }
    g->sweeps_done += nsweeps;
    return NSK_OK;
}

extern "C" {

int nsk_learn_sweeps(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                     double reg_param, int64_t truncation, int learn_non_evidence) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    if (regularization == 1 && truncation == 0) return fail(NSK_E_INVALID, "truncation must be non-zero (ZeroDivisionError in the reference)");
    if (nsweeps == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(g->device));
    return g->c.vbytes == 1
               ? learn_impl<int8_t>(g, nsweeps, step, decay, regularization, reg_param, truncation, learn_non_evidence)
               : learn_impl<int32_t>(g, nsweeps, step, decay, regularization, reg_param, truncation, learn_non_evidence);
}

int nsk_state_upload(nsk_graph *g, const int64_t *var_value, const int64_t *var_value_evid,
                     const double *weight_value, const int64_t *count) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    const size_t nvar = (size_t)g->c.nvar;
    const int64_t lo = g->c.vbytes == 1 ? -128 : INT32_MIN, hi = g->c.vbytes == 1 ? 127 : INT32_MAX;
    const int64_t *srcs[2] = {var_value, var_value_evid};
    void *dsts[2] = {g->val, g->val_evid};
    std::vector<int32_t> tmp;
    for (int k = 0; k < 2; k++) {
        if (!srcs[k]) continue;
        tmp.resize(nvar);
        for (size_t i = 0; i < nvar; i++) {
            if (srcs[k][i] < lo || srcs[k][i] > hi)
                return fail(NSK_E_RANGE, "variable value does not fit the device value type");
            tmp[i] = (int32_t)srcs[k][i];
        }
        int rc = upload_values(g, dsts[k], tmp.data(), nvar);
        if (rc) return rc;
    }
    if (weight_value && g->c.nweight) {
        HIPCHECK(hipMemcpyAsync(g->w, weight_value, (size_t)g->c.nweight * sizeof(double), hipMemcpyHostToDevice, g->stream));
        g->weights_dirty = true;
    }
    if (count && g->c.ncount) {
        HIPCHECK(hipMemcpyAsync(g->cnt_total, count, (size_t)g->c.ncount * sizeof(int64_t), hipMemcpyHostToDevice, g->stream));
        HIPCHECK(hipMemsetAsync(g->cnt, 0, (size_t)g->c.ncount * sizeof(int32_t), g->stream));
        if (g->c.npos) HIPCHECK(hipMemsetAsync(g->cnt_pos, 0, (size_t)g->c.npos, g->stream));
        g->pos_tally_sweeps = 0;
        g->cnt_dirty = false;
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

static int download_values(nsk_graph *g, const void *src, int64_t *dst) {
    const size_t nvar = (size_t)g->c.nvar;
    if (g->c.vbytes == 1) {
        std::vector<int8_t> tmp(nvar);
        if (nvar) HIPCHECK(hipMemcpyAsync(tmp.data(), src, nvar, hipMemcpyDeviceToHost, g->stream));
        HIPCHECK(hipStreamSynchronize(g->stream));
        for (size_t i = 0; i < nvar; i++) dst[i] = tmp[i];
    } else {
        std::vector<int32_t> tmp(nvar);
        if (nvar) HIPCHECK(hipMemcpyAsync(tmp.data(), src, nvar * 4, hipMemcpyDeviceToHost, g->stream));
        HIPCHECK(hipStreamSynchronize(g->stream));
        for (size_t i = 0; i < nvar; i++) dst[i] = tmp[i];
    }
    return NSK_OK;
}

int nsk_state_download(nsk_graph *g, int64_t *var_value, int64_t *var_value_evid, double *weight_value,
                       int64_t *count) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    int rc;
    if (var_value && (rc = download_values(g, g->val, var_value))) return rc;
    if (var_value_evid && (rc = download_values(g, g->val_evid, var_value_evid))) return rc;
    if (weight_value && g->c.nweight)
        HIPCHECK(hipMemcpyAsync(weight_value, g->w, (size_t)g->c.nweight * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    if (count) {
        if ((rc = fold_counts(g))) return rc;
        if (g->c.ncount)
            HIPCHECK(hipMemcpyAsync(count, g->cnt_total, (size_t)g->c.ncount * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

static void fill_info(const Compiled &c, nsk_graph_info *info) {
    info->nvar = c.nvar;
    info->nowned = c.nsampled;
    info->ncolors = (int64_t)c.phase_start.size() - 1;
    info->value_bytes = c.vbytes;
    info->device_bytes = 0;
    info->nfast = c.nfast;
    info->ngeneric = c.nsampled - c.nfast;
    info->alg_bytes_inference = c.alg_bytes_inference;
    info->alg_bytes_learning = c.alg_bytes_learning;
    info->sweeps_done = 0;
}

int nsk_graph_get_info(nsk_graph *g, nsk_graph_info *info) {
    if (!g || !info) return fail(NSK_E_INVALID, "null argument");
    fill_info(g->c, info);
    info->device_bytes = g->device_bytes;
    info->sweeps_done = g->sweeps_done;
    return NSK_OK;
}

int nsk_graph_plan(const nsk_graph_desc *desc, int32_t *color, nsk_graph_info *info) {
    if (!desc) return fail(NSK_E_INVALID, "null argument");
    Compiled c;
    std::string err;
    int rc = compile_graph(desc, c, err);
    if (rc) return fail(rc, err);
    if (color && c.nvar) memcpy(color, c.color.data(), (size_t)c.nvar * sizeof(int32_t));
    if (info) fill_info(c, info);
    return NSK_OK;
}

int nsk_graph_plan_needs(const nsk_graph_desc *desc, int64_t *count, int32_t *vids) {
    if (!desc || !count) return fail(NSK_E_INVALID, "null argument");
    Compiled c;
    std::string err;
    int rc = compile_graph(desc, c, err);
    if (rc) return fail(rc, err);
    *count = (int64_t)c.ghost_needs.size();
    if (vids && *count) memcpy(vids, c.ghost_needs.data(), (size_t)*count * sizeof(int32_t));
    return NSK_OK;
}

int nsk_graph_get_colors(nsk_graph *g, int32_t *color) {
    if (!g || !color) return fail(NSK_E_INVALID, "null argument");
    if (g->c.nvar) memcpy(color, g->c.color.data(), (size_t)g->c.nvar * sizeof(int32_t));
    return NSK_OK;
}

int nsk_profile_begin(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    g->launches_at_begin = g->launches;
    HIPCHECK(hipEventRecord(g->ev0, g->stream));
    return NSK_OK;
}

int nsk_profile_end(nsk_graph *g, double *elapsed_ms, int64_t *kernel_launches) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipEventRecord(g->ev1, g->stream));
    HIPCHECK(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    if (kernel_launches) *kernel_launches = g->launches - g->launches_at_begin;
    return NSK_OK;
}

int nsk_device_buffer(nsk_graph *g, int which, void **ptr, int64_t *nbytes) {
    if (!g || !ptr) return fail(NSK_E_INVALID, "null argument");
    switch (which) {
    case NSK_BUF_VALUE: *ptr = g->val; if (nbytes) *nbytes = g->c.nvar * g->c.vbytes; return NSK_OK;
    case NSK_BUF_VALUE_EVID: *ptr = g->val_evid; if (nbytes) *nbytes = g->c.nvar * g->c.vbytes; return NSK_OK;
    case NSK_BUF_WEIGHT: *ptr = g->w; if (nbytes) *nbytes = g->c.nweight * 8; g->weights_exposed = true; return NSK_OK;
    case NSK_BUF_SEND: *ptr = g->x_send; if (nbytes) *nbytes = g->xslot * g->c.vbytes; return NSK_OK;
    case NSK_BUF_RECV: *ptr = g->x_recv; if (nbytes) *nbytes = g->xslot * g->c.vbytes * g->xworld; return NSK_OK;
    case NSK_BUF_SEND_EVID: *ptr = g->x_send_evid; if (nbytes) *nbytes = g->xslot * g->c.vbytes; return NSK_OK;
    case NSK_BUF_RECV_EVID: *ptr = g->x_recv_evid; if (nbytes) *nbytes = g->xslot * g->c.vbytes * g->xworld; return NSK_OK;
    default: return fail(NSK_E_INVALID, "unknown buffer id");
    }
}

int nsk_selftest_exp(int device, const double *x, double *y, int64_t n) {
    if (n < 0 || (n && (!x || !y))) return fail(NSK_E_INVALID, "bad argument");
    if (n == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    HIPCHECK(hipMalloc((void **)&dx, n * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dy, n * sizeof(double)));
    HIPCHECK(hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice));
    k_selftest_exp<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(dx, dy, n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(y, dy, n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return NSK_OK;
}

int nsk_selftest_philox(int device, uint64_t seed, uint64_t sweep, uint32_t stream, int64_t n,
                        uint32_t *out) {
    if (n < 0 || (n && !out)) return fail(NSK_E_INVALID, "bad argument");
    if (n == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(device));
    uint32_t *d = nullptr;
    HIPCHECK(hipMalloc((void **)&d, 4 * n * sizeof(uint32_t)));
    k_selftest_philox<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(
        (uint32_t)seed, (uint32_t)(seed >> 32), stream, (uint32_t)sweep, (uint32_t)(sweep >> 32), n, d);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(out, d, 4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return NSK_OK;
}

int nsk_selftest_stream(int device, int64_t nbytes, int width, int iters, double *gbytes_per_s) {
    if (nbytes < 4096 || (width != 4 && width != 16) || iters < 1 || !gbytes_per_s) return fail(NSK_E_INVALID, "bad argument");
    HIPCHECK(hipSetDevice(device));
    nbytes &= ~(int64_t)4095;
    void *a = nullptr, *b = nullptr;
    HIPCHECK(hipMalloc(&a, nbytes));
    HIPCHECK(hipMalloc(&b, nbytes));
    HIPCHECK(hipMemset(a, 1, nbytes));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    const long long n = nbytes / width;
    const int grid = 256 * 8;
    for (int it = -1; it < iters; it++) {
        if (it == 0) HIPCHECK(hipEventRecord(e0, 0));
        if (width == 4) k_stream_copy<uint32_t><<<dim3(grid), dim3(NSK_BLOCK)>>>((const uint32_t *)a, (uint32_t *)b, n);
        else k_stream_copy<uint4><<<dim3(grid), dim3(NSK_BLOCK)>>>((const uint4 *)a, (uint4 *)b, n);
    }
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    *gbytes_per_s = 2.0 * (double)nbytes * iters / ((double)ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    return NSK_OK;
}

int nsk_ghost_needs(nsk_graph *g, int64_t *count, int32_t *vids) {
    if (!g || !count) return fail(NSK_E_INVALID, "null argument");
    *count = (int64_t)g->c.ghost_needs.size();
    if (vids && *count) memcpy(vids, g->c.ghost_needs.data(), (size_t)*count * sizeof(int32_t));
    return NSK_OK;
}

int nsk_exchange_setup(nsk_graph *g, int world, int rank, const int32_t *send_vids, int64_t nsend,
                       const int32_t *recv_vids, const int64_t *recv_off, int64_t slot) {
    if (!g || world < 1 || rank < 0 || rank >= world || nsend < 0 || slot < nsend || !recv_off)
        return fail(NSK_E_INVALID, "bad exchange description");
    HIPCHECK(hipSetDevice(g->device));
    const int64_t nrecv = recv_off[world];
    for (int64_t i = 0; i < nsend; i++)
        if (send_vids[i] < g->c.own_begin || send_vids[i] >= g->c.own_end)
            return fail(NSK_E_INDEX, "send list names a variable this handle does not own");
    std::vector<int32_t> rslot((size_t)nrecv);
    for (int src = 0; src < world; src++) {
        if (recv_off[src + 1] - recv_off[src] > slot) return fail(NSK_E_INVALID, "slot smaller than a rank's list");
        for (int64_t j = recv_off[src]; j < recv_off[src + 1]; j++) {
            if (recv_vids[j] < 0 || recv_vids[j] >= g->c.nvar) return fail(NSK_E_INDEX, "receive list out of range");
            rslot[j] = src == rank ? -1 : (int32_t)(src * slot + (j - recv_off[src]));
        }
    }
    g->xworld = world; g->xrank = rank; g->xslot = slot; g->xnsend = nsend; g->xnrecv = nrecv;
    std::vector<int32_t> sv(send_vids, send_vids + nsend), rv(recv_vids, recv_vids + nrecv);
    int rc;
    if ((rc = dev_upload(g, &g->x_send_vids, sv))) return rc;
    if ((rc = dev_upload(g, &g->x_recv_vids, rv))) return rc;
    if ((rc = dev_upload(g, &g->x_recv_slot, rslot))) return rc;
    const size_t vb = (size_t)g->c.vbytes;
    uint8_t *t = nullptr;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb))) return rc; g->x_send = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb * world))) return rc; g->x_recv = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb))) return rc; g->x_send_evid = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb * world))) return rc; g->x_recv_evid = t;
    HIPCHECK(hipMemsetAsync(g->x_send, 0, (size_t)(slot ? slot : 1) * vb, g->stream));
    HIPCHECK(hipMemsetAsync(g->x_send_evid, 0, (size_t)(slot ? slot : 1) * vb, g->stream));
    if ((rc = dev_alloc(g, &g->w_start, (size_t)g->c.nweight))) return rc;
    if ((rc = dev_alloc(g, &g->w_delta, (size_t)g->c.nweight))) return rc;
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

}  // extern "C"

template <typename VT>
static int exchange_kernels(nsk_graph *g, int which, bool pack) {
    VT *val = (VT *)(which == NSK_BUF_VALUE ? g->val : g->val_evid);
    VT *sb = (VT *)(which == NSK_BUF_VALUE ? g->x_send : g->x_send_evid);
    VT *rb = (VT *)(which == NSK_BUF_VALUE ? g->x_recv : g->x_recv_evid);
    if (pack) {
        const int n = (int)g->xnsend;
        if (n > 0)
            k_exchange_pack<VT><<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, g->x_send_vids, sb, n);
    } else {
        const int n = (int)g->xnrecv;
        if (n > 0)
            k_exchange_unpack<VT><<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, g->x_recv_vids, g->x_recv_slot, rb, n);
    }
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

extern "C" {

static int exchange_step(nsk_graph *g, int which, bool pack) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup has not been called");
    if (which != NSK_BUF_VALUE && which != NSK_BUF_VALUE_EVID) return fail(NSK_E_INVALID, "bad buffer id");
    HIPCHECK(hipSetDevice(g->device));
    return g->c.vbytes == 1 ? exchange_kernels<int8_t>(g, which, pack) : exchange_kernels<int32_t>(g, which, pack);
}

int nsk_exchange_pack(nsk_graph *g, int which) { return exchange_step(g, which, true); }
int nsk_exchange_unpack(nsk_graph *g, int which) { return exchange_step(g, which, false); }

// ---- native RCCL loop -----------------------------------------------------------------------------
static int load_rccl(const char *path) {
    if (g_rccl.lib) return NSK_OK;
    void *h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(NSK_E_DEVICE, std::string("dlopen(librccl): ") + dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(NSK_E_DEVICE, "librccl lacks the expected symbols");
    g_rccl.lib = h;
    return NSK_OK;
}

#define RCCLCHECK(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(NSK_E_DEVICE, std::string(#expr) + ": " +                              \
                        (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"));    \
    } while (0)

int nsk_comm_unique_id(const char *librccl_path, void *id128) {
    if (!id128) return fail(NSK_E_INVALID, "null argument");
    int rc = load_rccl(librccl_path);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RCCLCHECK(g_rccl.GetUniqueId((ncclUniqueId *)id128));
    return NSK_OK;
}

int nsk_comm_init(nsk_graph *g, int world, int rank, const void *id128, const char *librccl_path) {
    if (!g || !id128 || world < 1 || rank < 0 || rank >= world) return fail(NSK_E_INVALID, "bad argument");
    int rc = load_rccl(librccl_path);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(g->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCLCHECK(g_rccl.CommInitRank(&comm, world, id, rank));
    g->rccl_comm = comm;
    return NSK_OK;
}

static int native_exchange(nsk_graph *g, int which) {
    int rc = exchange_step(g, which, true);
    if (rc) return rc;
    const void *sb = which == NSK_BUF_VALUE ? g->x_send : g->x_send_evid;
    void *rb = which == NSK_BUF_VALUE ? g->x_recv : g->x_recv_evid;
    if (g->xslot > 0)
        RCCLCHECK(g_rccl.AllGather(sb, rb, (size_t)g->xslot, g->c.vbytes == 1 ? ncclInt8 : ncclInt32,
                                   (ncclComm_t)g->rccl_comm, g->stream));
    return exchange_step(g, which, false);
}

int nsk_gibbs_sweeps_exchange(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->rccl_comm || g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup / nsk_comm_init first");
    for (int64_t s = 0; s < nsweeps; s++) {
        int rc = nsk_gibbs_sweeps(g, 1, sample_evidence, burnin);
        if (rc) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE))) return rc;
    }
    return NSK_OK;
}

int nsk_learn_sweeps_exchange(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                              double reg_param, int64_t truncation, int learn_non_evidence) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->rccl_comm || g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup / nsk_comm_init first");
    const int nw = (int)g->c.nweight;
    for (int64_t s = 0; s < nsweeps; s++) {
        HIPCHECK(hipSetDevice(g->device));
        if (nw) HIPCHECK(hipMemcpyAsync(g->w_start, g->w, (size_t)nw * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        int rc = nsk_learn_sweeps(g, 1, step, 1.0, regularization, reg_param, truncation, learn_non_evidence);
        if (rc) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE))) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE_EVID))) return rc;
        if (nw) {       // w = w_start + sum over ranks of (w - w_start): numbskull_master.py:223-224
            const dim3 grid((nw + NSK_BLOCK - 1) / NSK_BLOCK), block(NSK_BLOCK);
            k_weight_delta<<<grid, block, 0, g->stream>>>(g->w, g->w_start, g->w_delta, nw);
            RCCLCHECK(g_rccl.AllReduce(g->w_delta, g->w_delta, (size_t)nw, ncclDouble, ncclSum,
                                       (ncclComm_t)g->rccl_comm, g->stream));
            k_weight_merge<<<grid, block, 0, g->stream>>>(g->w, g->w_start, g->w_delta, nw);
            g->weights_dirty = true;
        }
        step *= decay;
    }
    return NSK_OK;
}

}  // extern "C"
