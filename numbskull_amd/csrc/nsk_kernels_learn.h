// nsk_kernels_learn.h -- learning sweep kernels (learnthread / sample_and_sgd,
// numbskull/learning.py:12-125) and the per-class weight update.
#pragma once

#include "nsk_device.h"
#include "nsk_kernels_gibbs.h"

namespace nsk {

// ---------------------------------------------------------------------------------------------
// Learning.  One colour class of one learning sweep = sample_and_sgd (learning.py:46-125) for every
// variable of the class with the weights frozen; gradients go to per-weight fixed-point sums and
// the weight update for the whole class is applied afterwards (DESIGN.md "device-mode learning").
// ---------------------------------------------------------------------------------------------
struct LearnParams {
    int regularization, learn_non_evidence;
    double inv_trunc;
    uint32_t k0, k1, s0, s1;
    int hub0;                   // first hub descriptor of the colour class
    int kstat;                  // the weight update adds the structural visit counts (nsk_compile.h ep_kstat):
                                // entries of dataType-0 variables with a zero gradient skip the accumulators
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
        // this XCD's private copy (HW_REG_XCC_ID, bits 3:0): the adds then stay in its own L2
        const uint32_t xcc = g.acc_copies > 1
            ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (NSK_XCDS - 1)) : 0u;
        const size_t off = (size_t)xcc * (size_t)g.nweight;
        sk.G = g.G + off; sk.K = g.K + off; sk.T = g.T + off;
    }
    sk.local = SMALLW || g.acc_copies > 1;
    sk.packed = !SMALLW && g.packed_grad != 0;
    sk.w_direct = SMALLW ? nullptr : g.w_direct;
    sk.w = g.w;
    sk.step = g.upd_step; sk.reg_param = g.upd_reg_param; sk.truncation = g.upd_truncation; sk.cap = g.upd_cap;
    sk.grad_inv = g.grad_inv; sk.regularization = g.upd_regularization; sk.clipped = g.upd_clipped; sk.a1 = g.upd_a1;
    return sk;
}

// SMALLW: a block's LDS sums go to one of NSK_LEARN_BINS bins per weight with global atomics
// (integer sums: order-free); k_apply_bins adds the bins up.  Spreading the blocks over the bins
// keeps the same-address atomic chains short, and there is no per-block row to budget for.
// bins_xcd (gfx942 / gfx950, as the XCD-private accumulators of open_sink): the bins are dealt to
// the XCDs, eight each, and a block adds to the bins of the XCD it runs on (HW_REG_XCC_ID) with
// workgroup-scope atomics, which execute in that XCD's L2.  Agent-scope adds are carried out on the
// memory side of the fabric, and the 32 same-address adds of a resident grid's blocks -- all
// finishing together -- were a 3.6 us tail of the 25 us table launch (10M grid, LNOSINK ablation).
#define NSK_LEARN_BINS 64
template <bool SMALLW, typename VT>
__device__ __forceinline__ void close_sink(const DevGraph<VT> &g, const GradSink &sk) {
    if (!SMALLW) return;
    __syncthreads();
    const int nw = g.nweight;
    const bool own = g.bins_xcd != 0;
    const uint32_t xcc = own ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (NSK_XCDS - 1)) : 0u;
    static_assert(NSK_LEARN_BINS == 8 * NSK_XCDS, "eight bins per XCD");
    const size_t bin = (size_t)(own ? 8u * xcc + ((blockIdx.x >> 3) & 7u) : (blockIdx.x & (NSK_LEARN_BINS - 1))) * (size_t)nw;
    for (int i = (int)threadIdx.x; i < nw; i += NSK_BLOCK) {
        if (sk.K[i] == 0) continue;
        sink_add(own, (unsigned long long *)&g.part_G[bin + i], (unsigned long long)sk.G[i]);
        sink_add(own, &g.part_K[bin + i], sk.K[i]);
        if (sk.T[i]) sink_add(own, &g.part_T[bin + i], sk.T[i]);
    }
}

// Generic learning kernel.  Work items are 64-position groups: item i covers positions
// pbegin + 64 i (range mode) or list[i] (list mode: the per-lane-header tiles of the fast range),
// clipped at pend.  A persistent grid strides over the items so that SMALLW blocks flush once.
template <typename VT, bool SMALLW, bool INL>
__global__ __launch_bounds__(NSK_BLOCK, NSK_GENERIC_LEARN_WAVES) void k_learn_phase(DevGraph<VT> g, int pbegin, int pend,
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
        int self = 0, evidence = 0, proposal = 0, a = 0, ae = 0, b = 0, be = 0;
        const uint2 *ra = nullptr, *rb = nullptr;        // INL: cursors into the inline records
        if (p < pend && g.p_vid[p] >= 0) {
            const uint32_t info = g.p_info[p];
            const int ev = NSK_INFO_EV(info);
            const int slot0 = g.p_slot[p];
            const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
            if (ev != 1) evidence = draw_sample<VT, INL>(g, p, info, slot0, g.val_evid, u53(r.z, r.w));   // 54-58
            else evidence = (int)g.p_init[p];                                                            // 61-62
            g.val_evid[p] = (VT)evidence;
            proposal = draw_sample<VT, INL>(g, p, info, slot0, g.val, u53(r.x, r.y));                      // 66-70
            g.val[p] = (VT)proposal;
            self = p;
            if (lp.learn_non_evidence || ev == 1) {                                               // 71-72
                if (lp.regularization == 1) {                                                     // 90
                    const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
                    truncate = u53(t.x, t.y) < lp.inv_trunc;
                }
                const int step = NSK_INFO_DT1(info);
                const int sa = slot0 + step * evidence;
                a = g.slot_off[sa];
                ae = g.slot_off[sa + 1];
                if (INL) ra = g.gstream + g.gs_off[sa];
                if (step && evidence != proposal) {
                    b = g.slot_off[slot0 + proposal];
                    be = g.slot_off[slot0 + proposal + 1];
                    if (INL) rb = g.gstream + g.gs_off[slot0 + proposal];
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
                uint4 rec;
                const int2 *mb;
                double feat;
                if (INL) {
                    const int fa = a < ae ? (int)ra[1].y : 0x7fffffff;
                    const int fb = b < be ? (int)rb[1].y : 0x7fffffff;
                    const uint2 *r = fa <= fb ? ra : rb;
                    const uint2 h0 = r[0], h1 = r[1], h2 = r[2];
                    rec = uint4{h0.x, h1.x, h0.y, 0u};
                    mb = (const int2 *)(r + 4) - (int)h1.x;
                    feat = __longlong_as_double(((long long)h2.y << 32) | (long long)h2.x);
                    if (fa <= fb) { ra += 4 + ra[3].x; a++; }
                    if (fb <= fa) { rb += 4 + rb[3].x; b++; }
                } else {
                    const int fa = a < ae ? g.fidx[a] : 0x7fffffff;
                    const int fb = b < be ? g.fidx[b] : 0x7fffffff;
                    const int fid = fa < fb ? fa : fb;
                    if (fa == fid) a++;
                    if (fb == fid) b++;
                    rec = g.f_rec[fid];
                    mb = g.m_rec;
                    feat = g.f_feat[fid];
                }
                more = (a < ae) || (b < be);
                wid = (int)rec.z;
                if (!g.w_fixed[wid]) {                                                            // 100-101
                    const double p0 = eval_factor(g, rec, mb, self, evidence, g.val_evid);
                    const double p1 = eval_factor(g, rec, mb, self, proposal, g.val);
                    const double gradient = (p1 - p0) * feat;                                     // 109
                    gfix = __double2ll_rn(gradient * (double)g.grad_mul);
                    have = true;
                }
            }
            accumulate_gradient(sk, have, wid, gfix, truncate);
        }
    }
    close_sink<SMALLW>(g, sk);
}

// Learning for hubs: one wave per variable.  Draws use the wave-cooperative potentials; the
// gradient visits of list(evidence) and of the factors that are only in list(proposal) are spread
// over the lanes (membership of a proposal-list factor in the evidence list: binary search, both
// lists are sorted) and go through the same wave-aggregated accumulators.
template <typename VT>
__device__ __forceinline__ void learn_heavy_variable(const DevGraph<VT> &g, const GradSink &sk, int p,
                                                     const LearnParams &lp) {
    const int lane = (int)(threadIdx.x & 63);
    const int v = g.p_vid[p];
    if (v < 0) return;
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info);
    const int slot0 = g.p_slot[p];
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    int evidence;
    if (ev != 1) evidence = wave_draw_sample(g, p, info, slot0, g.val_evid, u53(r.z, r.w));
    else evidence = (int)g.p_init[p];
    const int proposal = wave_draw_sample(g, p, info, slot0, g.val, u53(r.x, r.y));
    if (lane == 0) { g.val_evid[p] = (VT)evidence; g.val[p] = (VT)proposal; }
    if (!(lp.learn_non_evidence || ev == 1)) return;
    bool truncate = false;
    if (lp.regularization == 1) {
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        truncate = u53(t.x, t.y) < lp.inv_trunc;
    }
    const int step = NSK_INFO_DT1(info);
    const int a = g.slot_off[slot0 + step * evidence], ae = g.slot_off[slot0 + step * evidence + 1];
    int b = 0, be = 0;
    if (step && evidence != proposal) { b = g.slot_off[slot0 + proposal]; be = g.slot_off[slot0 + proposal + 1]; }
    const int na = ae - a, nb = be - b;
    for (int base = 0; base < na + nb; base += 64) {                 // wave-uniform trip count
        const int i = base + lane;
        bool have = false;
        int wid = 0;
        long long gfix = 0;
        if (i < na + nb) {
            const int fid = i < na ? g.fidx[a + i] : g.fidx[b + (i - na)];
            bool dup = false;
            if (i >= na) {                                           // already visited via list(evidence)?
                int lo = a, hi = ae;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (g.fidx[mid] < fid) lo = mid + 1; else hi = mid; }
                dup = lo < ae && g.fidx[lo] == fid;
            }
            const uint4 rec = g.f_rec[fid];
            wid = (int)rec.z;
            if (!dup && !g.w_fixed[wid]) {
                const double p0 = eval_factor(g, rec, g.m_rec, p, evidence, g.val_evid);
                const double p1 = eval_factor(g, rec, g.m_rec, p, proposal, g.val);
                gfix = __double2ll_rn(((p1 - p0) * g.f_feat[fid]) * (double)g.grad_mul);
                have = true;
            }
        }
        accumulate_gradient(sk, have, wid, gfix, truncate);
    }
}

// Learning for an entry-parallel hub (heavy_update_ep's twin): both chains' potentials with one lane
// per entry, the draws, then a second pass over the entries for the gradient visits -- the union of
// list(evidence) and list(proposal) is "entries owned by every candidate, by the evidence value or by
// the proposal" (learning.py:76-95), each lane hands its entry to the wave-aggregated accumulators.
template <typename VT>
__device__ __forceinline__ void learn_heavy_variable_ep(const DevGraph<VT> &g, const GradSink &sk, const uint8_t *lut,
                                                        int p, const uint4 hd, const LearnParams &lp) {
    const int lane = (int)(threadIdx.x & 63);
    const int v = g.p_vid[p];
    if (v < 0) return;
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info), card = NSK_INFO_CARD(info);
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    int evidence;
    if (ev != 1) evidence = hub_draw(hub_potentials(g, lut, hd, g.val_evid), card, u53(r.z, r.w));   // 54-58
    else evidence = (int)g.p_init[p];                                                             // 61-62
    const int proposal = hub_draw(hub_potentials(g, lut, hd, g.val), card, u53(r.x, r.y));            // 66-70
    if (lane == 0) { g.val_evid[p] = (VT)evidence; g.val[p] = (VT)proposal; }
    if (!(lp.learn_non_evidence || ev == 1)) return;                                               // 71-72
    bool truncate = false;
    if (lp.regularization == 1) {                                                                  // 90
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        truncate = u53(t.x, t.y) < lp.inv_trunc;
    }
    const int n = (int)hd.y, M = (int)(hd.z & 0xFFu), rows = 2 + M;
    const uint32_t *base = g.hub_adj + hd.x;
    for (int rr = 0; rr * 64 < n; rr++) {
        HubEntry ef, ee;
        hub_entry(g, lut, base, rows, rr, M, g.val, false, ef);
        hub_entry(g, lut, base, rows, rr, M, g.val_evid, false, ee);
        const bool mine = entry_visited(ef.d1, evidence, proposal);
        const long long diff = (long long)(proposal == ef.cstar ? ef.A : ef.B) -
                               (long long)(evidence == ee.cstar ? ee.A : ee.B);
        const bool have = rr * 64 + lane < n && mine && !g.w_fixed[ef.wid];                        // 100-101
        accumulate_gradient(sk, have, (int)ef.wid, diff * g.grad_mul, truncate);
    }
}

// a hub position: entry-parallel when the graph compiler built its lane-per-entry stream
template <typename VT>
__device__ __forceinline__ void learn_hub(const DevGraph<VT> &g, const GradSink &sk, const uint8_t *lut, int p,
                                          int hubdesc, const LearnParams &lp) {
    const NSK_SCALAR uint32_t *hdp = (const NSK_SCALAR uint32_t *)(g.hub_desc + hubdesc);
    const uint4 hd = {hdp[0], hdp[1], hdp[2], hdp[3]};
    if (hd.w) return;                                      // a long list: a whole workgroup's (block_hub_learn)
    if (hd.y) learn_heavy_variable_ep<VT>(g, sk, lut, p, hd, lp);
    else learn_heavy_variable<VT>(g, sk, p, lp);
}

template <typename VT, bool SMALLW>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_heavy(DevGraph<VT> g, int pbegin, int pend,
                                                           LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) uint8_t lut[2048];
    load_gen_lut(lut);
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int wave0 = (int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
    const int nwaves = (int)(gridDim.x * (NSK_BLOCK / 64));
    for (int p = pbegin + wave0; p < pend; p += nwaves) learn_hub<VT>(g, sk, lut, p, lp.hub0 + (p - pbegin), lp);
    close_sink<SMALLW>(g, sk);
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
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
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
        g.val_evid[p] = (VT)evidence;
        g.val[p] = (VT)proposal;
    }
    const bool part = valid && (lp.learn_non_evidence || ev == 1);        // 71-72
    bool truncate = false;
    if (lp.regularization == 1) {                                         // 90
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
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
                const long long dG = (span * (long long)(nf - ne)) * g.grad_mul;     // Q31.32
                const int wid = (int)(s & 0xFFFFFFu);
                sink_add(sk.local, (unsigned long long *)&sk.G[wid], (unsigned long long)(dG + (sk.packed ? (long long)nk : 0)));
                if (!sk.packed) sink_add(sk.local, &sk.K[wid], nk);
                if (nt) sink_add(sk.local, &sk.T[wid], nt);
            }
        }
    }
}

// One shape tile of the learning sweep (per-lane functions and weights, shared word layout).
// Pass 1 walks the words for both chains, keeps the per-entry satisfied bits for candidates 0/1 in
// lane bitfields (entry e -> bit e) and accumulates the potentials; after the draws, pass 2
// re-reads the header words (cache hits) and adds each entry's gradient through the wave-aggregated
// accumulators, one entry position at a time (the role program makes the positions wave-uniform).
template <typename VT>
__device__ __forceinline__ void learn_tile_shape(const DevGraph<VT> &g, const GradSink &sk, const uint4 *sp,
                                                 int len, uint32_t prog, int p, bool valid,
                                                 const LearnParams &lp, double *wsave = nullptr) {
    // wsave: LDS, 8 x NSK_BLOCK doubles or null -- pass 1 leaves the weights of a lane's first 8 entries at
    // [entry][thread], so pass 2 updates a single-factor weight in place without gathering it again
    const NSK_SCALAR uint32_t *rp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + prog);
    const uint32_t info = valid ? g.p_info[p] : 0u;
    const int ev = NSK_INFO_EV(info);
    const int init = valid ? (int)g.p_init[p] : 0;
    const bool need_evid = __ballot(valid && ev != 1) != 0;

    double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
    uint32_t B0 = 0, B1 = 0, C0 = 0, C1 = 0;
    uint32_t code = 0;
    double w = 0.0;
    int ff = 0, fe = 0;                                        // first member value, free / evidence chain
    bool nzf = true, onef = false, eqf = true, nze = true, onee = false, eqe = true;
    int entry = -1;                                            // wave-uniform entry counter
    auto close = [&](bool nomember) {
        const bool isEq = code == 4u, isAnd = code == 3u || code == 1u, isOr = code == 2u;
        const bool b0 = (isEq && eqf && (nomember || ff == 0)) || (isOr && onef);
        const bool b1 = (isEq && eqf && (nomember || ff == 1)) || (isAnd && nzf) || isOr;
        const bool c0 = (isEq && eqe && (nomember || fe == 0)) || (isOr && onee);
        const bool c1 = (isEq && eqe && (nomember || fe == 1)) || (isAnd && nze) || isOr;
        const double hi = code == 0u ? 0.0 : 1.0;
        const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
        const double t0 = w * (b0 ? hi : lo), t1 = w * (b1 ? hi : lo);
        p0 = p0 + t0;
        p1 = p1 + t1;
        if (need_evid) {
            const double s0 = w * (c0 ? hi : lo), s1 = w * (c1 ? hi : lo);
            q0 = q0 + s0;
            q1 = q1 + s1;
        }
        B0 |= (b0 ? 1u : 0u) << entry; B1 |= (b1 ? 1u : 0u) << entry;
        C0 |= (c0 ? 1u : 0u) << entry; C1 |= (c1 ? 1u : 0u) << entry;
    };
    for (int c = 0; c * 4 < len; c++) {
        const uint4 q = sp[(size_t)c * 64];
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        uint32_t role[4];
#pragma unroll
        for (int i = 0; i < 4; i++) role[i] = rp[4 * c + i] & 0x1Fu;
        double wv[4];
        int xv[4], xe[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            wv[i] = 0.0; xv[i] = 0; xe[i] = 0;
            if (role[i] & 1u) wv[i] = g.w[wd[i] & 0xFFFFFFu];
            else if ((role[i] & 16u) && wd[i] != NSK_SHAPE_NULL) { xv[i] = (int)g.val[wd[i]]; xe[i] = (int)g.val_evid[wd[i]]; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (role[i] & 1u) {
                entry++;
                code = (0x343210u >> (4u * (wd[i] >> 27))) & 0xFu;
                w = wv[i];
                if (wsave && entry < 8) wsave[entry * NSK_BLOCK + (int)threadIdx.x] = w;
                ff = 0; fe = 0; nzf = true; onef = false; eqf = true; nze = true; onee = false; eqe = true;
                if (role[i] & 8u) close(true);
            } else if (role[i] & 16u) {
                const bool F = (role[i] & 2u) != 0;
                const bool nul = wd[i] == NSK_SHAPE_NULL;         // a slot the lane's entry lacks: the first member again
                const int x = nul ? ff : xv[i], y = nul ? fe : xe[i];
                eqf = F || (eqf && (x == ff)); nzf = (F || nzf) && (x != 0); onef = (!F && onef) || (x == 1);
                ff = F ? x : ff;
                eqe = F || (eqe && (y == fe)); nze = (F || nze) && (y != 0); onee = (!F && onee) || (y == 1);
                fe = F ? y : fe;
                if (role[i] & 4u) close(false);
            }
        }
    }
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    int evidence = init;
    if (need_evid && ev != 1) {
        const double z0 = nsk_exp(q0), z1 = z0 + nsk_exp(q1);
        const double z = u53(r.z, r.w) * z1;
        evidence = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    }
    const double z0 = nsk_exp(p0), z1 = z0 + nsk_exp(p1);
    const double z = u53(r.x, r.y) * z1;
    const int proposal = (z0 >= z) ? 0 : ((z1 >= z) ? 1 : 0);
    if (valid) {
        g.val_evid[p] = (VT)evidence;
        g.val[p] = (VT)proposal;
    }
    const bool part = valid && (lp.learn_non_evidence || ev == 1);
    bool truncate = false;
    if (lp.regularization == 1) {
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        truncate = part && (u53(t.x, t.y) < lp.inv_trunc);
    }
    if (__ballot(part) == 0) return;
    const uint32_t satf = proposal ? B1 : B0, sate = evidence ? C1 : C0;
    int e2 = -1;
    for (int c = 0; c * 4 < len; c++) {                         // pass 2: gradients, entry by entry
        uint32_t role[4];
#pragma unroll
        for (int i = 0; i < 4; i++) role[i] = rp[4 * c + i] & 0x1Fu;
        if (!((role[0] | role[1] | role[2] | role[3]) & 1u)) continue;      // no header in this chunk
        const uint4 q = sp[(size_t)c * 64];
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (role[i] & 1u) {
                e2++;
                const int wid = (int)(wd[i] & 0xFFFFFFu);
                const uint32_t cd = (0x343210u >> (4u * (wd[i] >> 27))) & 0xFu;
                const long long span = cd == 0u ? 0 : (cd == 1u ? 1 : 2);          // hi - lo
                // (a weight that is updated in place is never a fixed one: its flag is not even looked up)
                const bool dir = part && sk.w_direct && ((sk.w_direct[(uint32_t)wid >> 5] >> ((uint32_t)wid & 31u)) & 1u);
                const bool have = part && (dir || !g.w_fixed[wid]);
                const long long diff = (long long)((satf >> e2) & 1u) - (long long)((sate >> e2) & 1u);
                const bool saved = wsave != nullptr && e2 < 8;
                accumulate_gradient(sk, have, wid, (span * diff) * g.grad_mul, truncate, true, saved,
                                    saved ? wsave[e2 * NSK_BLOCK + (int)threadIdx.x] : 0.0);
            }
        }
    }
}

// One general tile (kind 6) of the learning sweep: pass 1 walks both chains for the potentials,
// the draws follow, pass 2 walks again (cache hits) and hands every entry's gradient
// value(proposal | free chain) - value(evidence | evidence chain) to the wave-aggregated
// accumulators.  An entry is visited when it belongs to every candidate (dataType 0) or to the
// evidence or the proposal value -- the union of the two factor lists of learning.py:76-95 (an
// entry is in one list only: variables whose own edges in one factor disagree on dense_equal_to
// stay on the generic path).
// `facts`: this wave's LDS array of NSK_LEARN_FACTS x 64 uint16.  When pass 1 walks both chains it
// leaves every entry's closed facts there -- (cstar clamped to 15, A + 1, B + 1) per chain, 8 bits
// each -- and pass 2 only re-reads the entries' weight ids and descriptors from the stream: no
// second round of member gathers, chain updates and table look-ups.
#define NSK_LEARN_FACTS 32
__device__ __forceinline__ uint32_t pack_facts(int cstar, int A, int B) {
    return (uint32_t)(cstar > 15 ? 15 : cstar) | ((uint32_t)(A + 1) << 4) | ((uint32_t)(B + 1) << 6);
}
template <typename VT, int MAXC>
__device__ __forceinline__ void learn_tile_general(const DevGraph<VT> &g, const GradSink &sk, const uint8_t *lut,
                                                   const uint4 *sp, uint32_t tdw, uint32_t prog, int p,
                                                   bool valid, const LearnParams &lp, uint16_t *facts) {
    const int len = (int)(tdw & 0xFFu), maxcard = (int)((tdw >> 12) & 15u);
    const uint32_t info = valid ? g.p_info[p] : (2u << 9);
    const int ev = NSK_INFO_EV(info), card = NSK_INFO_CARD(info);
    const bool need_evid = __ballot(valid && ev != 1) != 0;
    const int Mslots = (int)((tdw >> 16) & 7u);
    const int lane = (int)(threadIdx.x & 63);
    // pass 2 from the saved facts: both chains walked in pass 1, a layout general_walk_ids knows, and room
    const bool saved = need_evid && Mslots <= 3 && len / (2 + Mslots) <= NSK_LEARN_FACTS;    // wave-uniform
    GenPot<MAXC> pf, pe;
    pf.clear(); pe.clear();
    if (need_evid) {
        int ei = 0;                                                      // wave-uniform entry counter
        general_walk<VT, true, 0, false>(g, g.val, g.val_evid, sp, len, Mslots, prog, nullptr,
                                  [&](uint32_t, double w, uint32_t d1, const GenChain &a, const GenChain &b) {
                                      int cstar, A, B;
                                      a.close(d1, lut, cstar, A, B);
                                      pf.add(maxcard, d1, w, cstar, A, B);
                                      uint32_t fx = pack_facts(cstar, A, B);
                                      b.close(d1, lut, cstar, A, B);
                                      pe.add(maxcard, d1, w, cstar, A, B);
                                      fx |= pack_facts(cstar, A, B) << 8;
                                      if (saved) facts[ei * 64 + lane] = (uint16_t)fx;
                                      ei++;
                                  });
    } else
        general_walk<VT, false, 0, false>(g, g.val, g.val, sp, len, (int)((tdw >> 16) & 7u), prog, nullptr,
                                   [&](uint32_t, double w, uint32_t d1, const GenChain &a, const GenChain &) {
                                       int cstar, A, B;
                                       a.close(d1, lut, cstar, A, B);
                                       pf.add(maxcard, d1, w, cstar, A, B);
                                   });
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    int evidence = valid ? (int)g.p_init[p] : 0;                                       // learning.py:61-62
    if (need_evid && ev != 1) evidence = pe.draw(maxcard, card, u53(r.z, r.w));        // 54-58
    const int proposal = pf.draw(maxcard, card, u53(r.x, r.y));                        // 66-70
    if (valid) {
        g.val_evid[p] = (VT)evidence;
        g.val[p] = (VT)proposal;
    }
    const bool part = valid && (lp.learn_non_evidence || ev == 1);                     // 71-72
    bool truncate = false;
    if (lp.regularization == 1) {
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        truncate = part && (u53(t.x, t.y) < lp.inv_trunc);
    }
    if (__ballot(part) == 0) return;
    if (saved) {
        general_walk_ids(sp, len, Mslots, [&](int e, uint32_t wid, uint32_t d1) {
            const bool mine = entry_visited(d1, evidence, proposal);
            const uint32_t fx = facts[e * 64 + lane];
            const int cf = (int)(fx & 15u), Af = (int)((fx >> 4) & 3u) - 1, Bf = (int)((fx >> 6) & 3u) - 1;
            const int ce = (int)((fx >> 8) & 15u), Ae = (int)((fx >> 12) & 3u) - 1, Be = (int)((fx >> 14) & 3u) - 1;
            const long long diff = (long long)(proposal == cf ? Af : Bf) - (long long)(evidence == ce ? Ae : Be);
            const bool have = part && mine && !g.w_fixed[wid];          // 100-101
            accumulate_gradient(sk, have, (int)wid, diff * g.grad_mul, truncate);
        });
        return;
    }
    general_walk<VT, true, 2, false>(g, g.val, g.val_evid, sp, len, (int)((tdw >> 16) & 7u), prog, nullptr,
                           [&](uint32_t wid, double, uint32_t d1, const GenChain &a, const GenChain &b) {
                               const bool mine = entry_visited(d1, evidence, proposal);
                               int cf, Af, Bf, ce, Ae, Be;
                               a.close(d1, lut, cf, Af, Bf);
                               b.close(d1, lut, ce, Ae, Be);
                               const long long diff = (long long)(proposal == cf ? Af : Bf) -
                                                      (long long)(evidence == ce ? Ae : Be);
                               const bool have = part && mine && !g.w_fixed[wid];      // 100-101
                               accumulate_gradient(sk, have, (int)wid, diff * g.grad_mul, truncate);
                           });
}

// One uniform / shape tile (tile t of the colour) of the learning sweep, descriptor-driven
template <typename VT>
__device__ __forceinline__ void learn_rest_tile(const DevGraph<VT> &g, const GradSink &sk, int pbegin, int pend,
                                                int wb_base, int t, const LearnParams &lp, double *wsave = nullptr) {
    const int lane = (int)(threadIdx.x & 63);
    const NSK_SCALAR uint32_t *tdp = (const NSK_SCALAR uint32_t *)(g.tiles + (wb_base + t));
    const struct { uint32_t x, y, z, w; } td = {tdp[0], tdp[1], tdp[2], tdp[3]};
    if (td.z == NSK_PAD_WORD) return;                        // mixed per-lane headers: generic kernel's job
    const int p = pbegin + t * 64 + lane;
    const bool valid = p < pend && g.p_vid[p] >= 0;
    const uint4 *sp = g.adj + td.x + lane;
    const uint32_t kind = (td.w >> 8) & 7u;
    if (kind == 7u) learn_tile_shape<VT>(g, sk, sp, (int)(td.w & 0xFFu), td.z, p, valid, lp, wsave);
    else if (kind == 4u) learn_tile<VT, 4>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
    else if (kind == 0u) learn_tile<VT, 0>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
    else if (kind == 2u) learn_tile<VT, 2>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
    else learn_tile<VT, 3>(g, sk, sp, (int)td.y, td.z, p, valid, lp);
}

// Learning over the uniform and shape tiles of a colour class that are not in a segment launch
// (tiles with per-lane headers are left to k_learn_phase in list mode).  Each wave takes a
// contiguous run of the list.
// Learning over homogeneous segments (runs of uniform tiles with one program; see k_gibbs_seg): the
// tile kind and chunk count are template parameters, so the body is the one learn_tile variant the
// segments need, with every loop bound known at compile time.
template <typename VT, bool SMALLW, int KIND, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_seg(DevGraph<VT> g, SegTable tab, LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6)));
    const int nwaves = (int)(gridDim.x * (NSK_BLOCK / 64));
    const int ntiles = tab.ntiles;
    const int per = (ntiles + nwaves - 1) / nwaves;
    const int t1 = min(ntiles, (wave0 + 1) * per);
    for (int T = wave0 * per; T < t1; T++) {
        int sidx = 0;
#pragma unroll
        for (int i = 1; i < NSK_SEG_MAX; i++) sidx += (i < tab.n && T >= tab.e[i].tile_start) ? 1 : 0;
        const SegEntry en = tab.e[sidx];
        const int t = T - en.tile_start;
        const int p = en.pos0 + t * 64 + lane;
        const bool valid = g.p_vid[p] >= 0;                  // -1: padding lane at a class end
        const uint4 *sp = g.adj + en.adj_off + (size_t)t * (64 * NCH) + lane;
        learn_tile<VT, KIND>(g, sk, sp, 4 * NCH, en.prog, p, valid, lp);
    }
    close_sink<SMALLW>(g, sk);
}

// SMALLW: one block adds up the bins of every weight (and clears them), applies the update,
// rewrites prog_w and rebuilds the draw tables from the new weights (the same products prog_w
// holds) -- one small launch per colour class.  (Running it as the tail of the class's learning
// launch -- last block to finish, found with a ticket counter -- was tried: 2048 same-address
// agent-scope atomics serialise at ~0.125 us each, 262 us per class instead of 33.)
#define NSK_APPLY_STAGE_WORDS 2048
#define NSK_APPLY_STAGE_PROGS 64
struct ApplyArgs {
    double *w;                  // weights after the update (every weight is written)
    const double *w_in;         // weights before it (another array under the one-class lag, nsk_set_learn_lag)
    long long *part_G;
    uint32_t *part_K, *part_T;
    int nweight;
    double step;
    int regularization;
    double reg_param, truncation;
    const uint32_t *prog;
    double *prog_w;
    int nprog;
    const ZProgDev *zp;
    int nzp, nztab;
    uint4 *ztab;
    double cap;
    unsigned int *clipped;
    double grad_inv;            // 2^-(fraction bits of the gradient sums)
};

// every thread of the launch's ONE block calls it
__device__ __forceinline__ void apply_bins_block(const ApplyArgs &aa) {
    __shared__ double sw[NSK_SMALLW];
    // the slot programs and table descriptors are staged in LDS while the bins arrive: the later
    // phases would otherwise start with chains of dependent global loads (this is a pure latency
    // chain on the critical path between two colour classes)
    __shared__ uint32_t sprog[NSK_APPLY_STAGE_WORDS];
    __shared__ ZProgDev szp[NSK_APPLY_STAGE_PROGS];
    static_assert(NSK_LEARN_BINS == 64, "one bin per lane");
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int nweight = aa.nweight, nprog = aa.nprog, nzp = aa.nzp, nztab = aa.nztab;
    const uint32_t *prog = aa.prog;
    const ZProgDev *zp = aa.zp;
    const bool staged = nprog <= NSK_APPLY_STAGE_WORDS && nzp <= NSK_APPLY_STAGE_PROGS;
    if (staged) {
        for (int j = tid; j < nprog; j += NSK_BLOCK) sprog[j] = prog[j];
        for (int z = tid; z < nzp; z += NSK_BLOCK) szp[z] = zp[z];
        prog = sprog;
        zp = szp;
    }
    // wave k sums weights k, k + 4, ...: lane b reads (and clears) bin b, the wave adds the lanes up
    for (int i = tid >> 6; i < nweight; i += NSK_BLOCK / 64) {
        const size_t at = (size_t)lane * nweight + i;
        long long G = aa.part_G[at];
        long long K = (long long)aa.part_K[at], T = (long long)aa.part_T[at];
        if (K) { aa.part_G[at] = 0; aa.part_K[at] = 0; aa.part_T[at] = 0; }
        G = wave_sum_i64(G); K = wave_sum_i64(K); T = wave_sum_i64(T);
        if (lane == 0) {
            double x = aa.w_in[i];
            if (K > 0)
                x = apply_update(x, G, (uint32_t)K, (uint32_t)T, aa.step, aa.regularization, aa.reg_param,
                                 aa.truncation, aa.cap, aa.clipped, aa.grad_inv);
            aa.w[i] = x;
            sw[i] = x;
        }
    }
    __syncthreads();
    for (int j = tid; j < nprog; j += NSK_BLOCK) {
        const uint32_t s = prog[j];
        if (s >> 31) continue;
        const uint32_t code = (s >> 24) & 7u, wid = s & 0xFFFFFFu;
        if ((int)wid >= nweight) continue;
        const bool last = (s >> 28) & 1u;
        const double hi = code == 0u ? 0.0 : 1.0;
        const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
        const double x = sw[wid];
        aa.prog_w[2 * j] = last ? x * hi : 0.0;
        aa.prog_w[2 * j + 1] = last ? x * lo : 0.0;
    }
    // all table entries of all programs, flattened over the threads
    for (int e = tid; e < nztab; e += NSK_BLOCK) {
        int z = 0;
        while (z + 1 < nzp && (uint32_t)e >= zp[z + 1].off) z++;
        const ZProgDev zz = zp[z];
        aa.ztab[e] = ztab_entry(prog + zz.prog, zz.nslots, (uint32_t)e - zz.off,
                                [&](uint32_t, uint32_t s, double &thi, double &tlo) {
                                    const uint32_t code = (s >> 24) & 7u;
                                    const bool last = (s >> 28) & 1u;
                                    const double hi = code == 0u ? 0.0 : 1.0;
                                    const double lo = (code == 0u || code == 1u) ? 0.0 : -1.0;
                                    const double x = sw[s & 0xFFFFFFu];
                                    thi = last ? x * hi : 0.0;
                                    tlo = last ? x * lo : 0.0;
                                });
    }
}

static __global__ __launch_bounds__(NSK_BLOCK) void k_apply_bins(ApplyArgs aa) { apply_bins_block(aa); }

// The same over segments whose programs have draw tables (k_refresh_ztab): both chains' draws are
// integer compares against the tabulated thresholds, the per-slot satisfied bits come from the same
// table entries.  learn_tile's gradient bookkeeping, no float64 arithmetic.
//
// A "trip" is TPW consecutive tiles of one segment; the host numbers a launch's tiles virtually so
// that every segment is a whole number of trips (the dead tiles at a segment's end re-read its last
// tile and take no part).  The grid is resident (k_gibbs_seg_tab's XCD-contiguous ranges) and each
// wave runs a two-stage pipeline: the NEXT trip's member ids and evidence values are requested
// right after this trip's table entries, so the stream's HBM latency overlaps the draw, the stores
// and the gradient bookkeeping -- the wave is a chain of dependent round trips (ids -> member
// values -> table entries) and the ablations showed that chain, not instruction issue or bytes,
// bounding the launch.  No p_vid read: p_init is -1 at padding positions (nsk_compile.cpp).
// (Measured on the 10M grid, per class with the update launch: strided one-trip-per-visit grid 43 us,
// this kernel 32.5 us; requesting two trips ahead 35.3 us, forcing 8 waves per SIMD -- the kernel
// sits at 7 for its scalar registers -- 33.9 us.)
template <int NCH, int TPW>
struct LearnTrip {                       // what a trip needs from memory before its gathers
    uint32_t id[TPW][4 * NCH];
    int init[TPW];
};
struct LearnTripInfo {                   // wave-uniform
    int pos, nt, t0, ev;
    uint32_t prog, zoff, zmask;
};

// SMALLW launches carry NSK_SERVICE_BLOCKS extra blocks in front (one round of XCDs, so the tile blocks keep
// their XCDs): block 0 applies the weight update of the PREVIOUS colour class (`prev`; nweight = 0: none)
// while the other blocks sample this class -- under the one-class lag (nsk_set_learn_lag) nobody in this
// launch reads what it writes (the other weight set, that class's bins), and the next launch starts behind
// the kernel boundary.  The update is a 6 us single-block latency chain (bins -> update -> table
// thresholds) that three in-order fusions could not hide; here it is off the critical
// path by definition.
#define NSK_SERVICE_BLOCKS 8
template <typename VT, bool SMALLW, int NCH, int TPW>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_seg_tab(DevGraph<VT> g, SegTable tab, LearnParams lp, ApplyArgs prev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (SMALLW && blockIdx.x < NSK_SERVICE_BLOCKS) {                   // (block-uniform)
        if (blockIdx.x == 0 && prev.nweight > 0) apply_bins_block(prev);
        return;
    }
    const int bid = (int)blockIdx.x - (SMALLW ? NSK_SERVICE_BLOCKS : 0);
    const int gdim = (int)gridDim.x - (SMALLW ? NSK_SERVICE_BLOCKS : 0);
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int ntrips = tab.ntiles / TPW;
    const int per = (ntrips + 7) >> 3;                                  // trips per XCD
    const int xcd = bid & 7;
    const int wx = __builtin_amdgcn_readfirstlane((bid >> 3) * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    const int wpx = (gdim >> 3) * (NSK_BLOCK / 64);                     // waves per XCD (grid: multiple of 8)
    const int pend = min(ntrips, (xcd + 1) * per);
    // The wave keeps its gradient counts in scalar registers while the trips it walks share a slot
    // program: acc[j] = (satisfied under the proposal) - (satisfied under the evidence) of slot j's
    // factor summed over lanes and tiles, accK / accT the participating / truncating lanes.  They
    // reach the sink when the program changes and at the end -- the sums are integers, so the
    // grouping does not change them.
    uint32_t cur_prog = 0xFFFFFFFFu;
    int acc[4 * NCH];
    uint32_t accK = 0u, accT = 0u;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) acc[j] = 0;
    auto flush = [&]() {
        if (cur_prog != 0xFFFFFFFFu && accK != 0u) {
            const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + cur_prog);
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) {
                const uint32_t s = pp[j];
                const bool closes = (s >> 28) & 1u, fixed = (s >> 30) & 1u;      // uniform
                if (closes && !fixed && lane == 0) {
                    const uint32_t code = (s >> 24) & 7u;
                    const long long span = code == 0u ? 0 : (code == 1u ? 1 : 2);            // hi - lo
                    const long long dG = (span * (long long)acc[j]) * g.grad_mul;            // Q31.32
                    const int wid = (int)(s & 0xFFFFFFu);
                    sink_add(sk.local, (unsigned long long *)&sk.G[wid], (unsigned long long)(dG + (sk.packed ? (long long)accK : 0)));
                    if (!sk.packed) sink_add(sk.local, &sk.K[wid], accK);
                    if (accT) sink_add(sk.local, &sk.T[wid], accT);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4 * NCH; j++) acc[j] = 0;
        accK = 0u; accT = 0u;
    };
    // the segment of the last located trip stays in scalar registers: most launches have one
    int c_lo = 0, c_hi = 0, c_pos = 0, c_nt = 1;
    uint32_t c_adj = 0u, c_prog = 0u, c_zoff = 0u, c_zmask_ev = 0u, c_aff = NSK_NO_STREAM;
    auto issue = [&](int P, LearnTrip<NCH, TPW> &r, LearnTripInfo &ti) {
        const int T0 = P * TPW;
        if (T0 < c_lo || T0 >= c_hi) {                                  // wave-uniform, rare
            const int sidx = seg_of_tile(tab, T0);
            c_lo = tab.e[sidx].tile_start;
            c_hi = sidx + 1 < NSK_SEG_MAX ? tab.e[sidx + 1].tile_start : tab.ntiles;
            c_pos = tab.e[sidx].pos0; c_adj = tab.e[sidx].adj_off; c_prog = tab.e[sidx].prog;
            c_zoff = tab.e[sidx].zoff; c_zmask_ev = tab.e[sidx].zmask_ev; c_aff = tab.e[sidx].aff_off;
            c_nt = (int)(tab.e[sidx].ntiles_lead & 0x3FFFFFFFu);
        }
        ti.pos = c_pos; ti.nt = c_nt; ti.t0 = T0 - c_lo; ti.prog = c_prog; ti.zoff = c_zoff;
        ti.zmask = c_zmask_ev & 0xFFu; ti.ev = (int)(int8_t)(c_zmask_ev >> 8);
        // implicit adjacency (nsk_compile.h seg_aff): slot bases by scalar loads, member = base + lane;
        // the bases of all the trip's tiles are requested before the first is looked at
        uint32_t ab[TPW][4 * NCH];
#pragma unroll
        for (int k = 0; k < TPW; k++) {
            const int t = min(ti.t0 + k, c_nt - 1);                    // a dead tile re-reads the last one
            ab[k][0] = NSK_NO_STREAM;
            if (c_aff != NSK_NO_STREAM) {
                const NSK_SCALAR uint32_t *ap = (const NSK_SCALAR uint32_t *)(g.seg_aff + c_aff + (size_t)t * NCH);
#pragma unroll
                for (int j = 0; j < 4 * NCH; j++) ab[k][j] = ap[j];
            }
            r.init[k] = (int)g.p_init[c_pos + t * 64 + lane];          // -1: padding lane at a class end
        }
#pragma unroll
        for (int k = 0; k < TPW; k++) {
            const int t = min(ti.t0 + k, c_nt - 1);
            if (ab[k][0] != NSK_NO_STREAM) {                        // wave-uniform
#pragma unroll
                for (int j = 0; j < 4 * NCH; j++) r.id[k][j] = ab[k][j] + (uint32_t)lane;
            } else {
                const uint4 *sp = g.adj + c_adj + (size_t)t * (64 * NCH) + lane;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const uint4 q = sp[c * 64];
                    r.id[k][4 * c] = q.x; r.id[k][4 * c + 1] = q.y; r.id[k][4 * c + 2] = q.z; r.id[k][4 * c + 3] = q.w;
                }
            }
        }
    };
    LearnTrip<NCH, TPW> rn;
    LearnTripInfo in;
    int P = xcd * per + wx;
    for (; P < pend; P += wpx) {
        issue(P, rn, in);       // (with implicit adjacency a trip's requests are scalar loads and coalesced rows:
        const LearnTrip<NCH, TPW> r = rn;
        const LearnTripInfo ti = in;
        uint32_t idf[TPW], ide[TPW];
#pragma unroll
        for (int k = 0; k < TPW; k++) {
            uint32_t xf[4 * NCH], xe[4 * NCH];
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) {
                xf[j] = (uint32_t)(uint8_t)g.val[r.id[k][j]];
                xe[j] = (uint32_t)(uint8_t)g.val_evid[r.id[k][j]];
            }
            idf[k] = 0; ide[k] = 0;
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) { idf[k] |= xf[j] << j; ide[k] |= xe[j] << j; }    // (values are their bits: values_regular)
            idf[k] &= ti.zmask;
            ide[k] &= ti.zmask;
        }
        uint4 ef[TPW], ee[TPW];
#pragma unroll
        for (int k = 0; k < TPW; k++) { ef[k] = g.ztab[ti.zoff + idf[k]]; ee[k] = g.ztab[ti.zoff + ide[k]]; }
        // the next trip's requests go out behind the table loads (vmcnt counts in order: waiting
        // for the entries then leaves these in flight)
        if (ti.prog != cur_prog) { flush(); cur_prog = ti.prog; }             // uniform, rare
#pragma unroll
        for (int k = 0; k < TPW; k++) {
            const bool valid = ti.t0 + k < ti.nt && r.init[k] >= 0;
            const int p = ti.pos + (ti.t0 + k) * 64 + lane;
            const u32x4 rr = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
            int evidence = r.init[k];                                             // learning.py:61-62
            if (ti.ev != 1) evidence = k53(rr.z, rr.w) > ztab_K(ee[k]) ? 1 : 0;   // 54-58
            const int proposal = k53(rr.x, rr.y) > ztab_K(ef[k]) ? 1 : 0;         // 66-70
            if (valid) {
                g.val_evid[p] = (VT)evidence;
                g.val[p] = (VT)proposal;
            }
            const bool part = valid && (lp.learn_non_evidence || ti.ev == 1);     // 71-72
            if (lp.regularization == 1) {                                         // 90
                const u32x4 tt = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
                accT += (uint32_t)__popcll(__ballot(part && (u53(tt.x, tt.y) < lp.inv_trunc)));
            }
            accK += (uint32_t)__popcll(__ballot(part));
            // the slots' satisfied bits of the two chains, zero for lanes that do not take part
            const uint32_t satf = part ? (proposal ? (ef[k].z >> 8) : ef[k].z) : 0u;
            const uint32_t sate = part ? (evidence ? (ee[k].z >> 8) : ee[k].z) : 0u;
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++)
                acc[j] += __popcll(__ballot((satf >> j) & 1u)) - __popcll(__ballot((sate >> j) & 1u));
        }
    }
    flush();
    close_sink<SMALLW>(g, sk);
}

// ---- the learning launch of a class whose quads are (mostly) wide ones (nsk_compile.h seg_wide; int8 values) ----
// sample_and_sgd (learning.py:46-125) for four consecutive positions per lane: per member slot ONE dword load from each
// chain's value array brings the four neighbour bytes (k_gibbs_seg_tabw's trick), both chains' neighbourhoods are packed
// by shifts, the eight table entries come from the wave's LDS copy of the segment's draw table, both chains' new values
// leave as one dword each, and the gradient bookkeeping -- per slot: (satisfied under the proposal) - (satisfied under
// the evidence), summed over the class's visits -- stays in per-LANE counters (popcounts of packed bytes) until the slot
// program changes or the wave ends: no ballot, no cross-lane step per tile.  The generator is the learning sweeps' own:
// one block per POSITION, counter (position, stream, sweep) -- the oracle's learning mode is untouched.
// Quads that are not wide are sampled tile by tile (learn_tab_tile), as k_learn_seg_tab does.
struct TabwRest { uint32_t n, q[NSK_TABW_REST_MAX]; };             // the launch's quads that are not wide ones (TabwCold.rest)
__host__ __device__ inline int nsk_tabw_front(uint32_t n) { return (int)((n + 7u) & ~7u); }
template <int NCH>
struct WideLearnTrip { uint32_t xf[4 * NCH], xe[4 * NCH], init; };

// one tile of a table segment, tile by tile (the fall-back of the wide kernel): acc / accK / accT are the wave's scalar
// gradient counters
template <int NCH>
__device__ __forceinline__ void learn_tab_tile(const DevGraph<signed char> &g, const LearnParams &lp, const SegEntry &en, int t, int lane,
                                               int (&acc)[4 * NCH], uint32_t &accK, uint32_t &accT) {
    const int nt = (int)(en.ntiles_lead & 0x3FFFFFFFu);
    if (t < 0 || t >= nt) return;                                       // (wave-uniform: a dead tile of the quad)
    const uint32_t zmask = en.zmask_ev & 0xFFu;
    const int ev = (int)(int8_t)(en.zmask_ev >> 8);
    const int p = en.pos0 + t * 64 + lane;
    const int init = (int)g.p_init[p];                                  // -1: padding lane
    uint32_t id[4 * NCH];
    bool aff = false;
    if (en.aff_off != NSK_NO_STREAM) {
        const NSK_SCALAR uint32_t *ap = (const NSK_SCALAR uint32_t *)(g.seg_aff + en.aff_off + (size_t)t * NCH);
        aff = ap[0] != NSK_NO_STREAM;
        if (aff) {
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) id[j] = ap[j] + (uint32_t)lane;
        }
    }
    if (!aff) {
        const uint4 *sp = g.adj + en.adj_off + (size_t)t * (64 * NCH) + lane;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint4 q = sp[c * 64];
            id[4 * c] = q.x; id[4 * c + 1] = q.y; id[4 * c + 2] = q.z; id[4 * c + 3] = q.w;
        }
    }
    uint32_t idf = 0, ide = 0;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) {
        idf |= ((uint32_t)(uint8_t)g.val[id[j]] & 1u) << j;
        ide |= ((uint32_t)(uint8_t)g.val_evid[id[j]] & 1u) << j;
    }
    const uint4 ef = g.ztab[en.zoff + (idf & zmask)], ee = g.ztab[en.zoff + (ide & zmask)];
    const u32x4 rr = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    int evidence = init;                                                // learning.py:61-62
    if (ev != 1) evidence = k53(rr.z, rr.w) > ztab_K(ee) ? 1 : 0;       // 54-58
    const int proposal = k53(rr.x, rr.y) > ztab_K(ef) ? 1 : 0;          // 66-70
    const bool valid = init >= 0;
    if (valid) {
        g.val_evid[p] = (signed char)evidence;
        g.val[p] = (signed char)proposal;
    }
    const bool part = valid && (lp.learn_non_evidence || ev == 1);      // 71-72
    if (lp.regularization == 1) {                                       // 90
        const u32x4 tt = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        accT += (uint32_t)__popcll(__ballot(part && (u53(tt.x, tt.y) < lp.inv_trunc)));
    }
    accK += (uint32_t)__popcll(__ballot(part));
    const uint32_t satf = part ? (proposal ? (ef.z >> 8) : ef.z) : 0u;
    const uint32_t sate = part ? (evidence ? (ee.z >> 8) : ee.z) : 0u;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++)
        acc[j] += __popcll(__ballot((satf >> j) & 1u)) - __popcll(__ballot((sate >> j) & 1u));
}

template <bool SMALLW, int NCH>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_seg_tabw(DevGraph<signed char> g, SegTable tab, LearnParams lp, ApplyArgs prev, TabwRest rest) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (SMALLW && blockIdx.x < NSK_SERVICE_BLOCKS) {                   // (block-uniform) the previous class's update rides here
        if (blockIdx.x == 0 && prev.nweight > 0) apply_bins_block(prev);
        return;
    }
    constexpr int ST = NSK_WIDE_STRIDE(NCH), ZN = 1 << NSK_ZT_BITS(NCH), ZR = (ZN + 63) / 64;
    __shared__ uint4 zt_all[(NSK_BLOCK / 64) * ZN];                     // the wave's copy of its segment's table entries
    // behind the service blocks: a workgroup per quad that is not a wide one (TabwRest), a wave per tile -- see TabwCold.rest
    const int nfront = nsk_tabw_front(rest.n);
    const int bid0 = (int)blockIdx.x - (SMALLW ? NSK_SERVICE_BLOCKS : 0);
    if (bid0 < nfront && (uint32_t)bid0 >= rest.n) return;             // (block-uniform: a front workgroup without a quad)
    const int bid = bid0 - nfront;
    const int gdim = (int)gridDim.x - (SMALLW ? NSK_SERVICE_BLOCKS : 0) - nfront;
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 *zt = zt_all + wv * ZN;
    const int nquads = tab.ntiles >> 2;                                 // virtual tiles: every segment is whole quads
    const int per = (nquads + 7) >> 3;                                  // quads per XCD
    const int xcd = bid & 7;
    const int wx = __builtin_amdgcn_readfirstlane((bid >> 3) * (NSK_BLOCK / 64)) + wv;
    const int wpx = (gdim >> 3) * (NSK_BLOCK / 64);                     // waves per XCD (grid: multiple of 8)
    const int q0 = min(nquads, xcd * per), q1 = min(nquads, (xcd + 1) * per);
    // gradient counters: per lane while the quads are wide (accv: packed-byte popcounts), per wave for tiles (acc:
    // ballots); they reach the sink when the slot program changes and at the end -- integer sums, any grouping
    uint32_t cur_prog = 0xFFFFFFFFu;
    int acc[4 * NCH], accv[4 * NCH];
    uint32_t accK = 0u, accT = 0u, accKv = 0u, accTv = 0u;
#pragma unroll
    for (int j = 0; j < 4 * NCH; j++) { acc[j] = 0; accv[j] = 0; }
    auto flush = [&]() {
        if (cur_prog != 0xFFFFFFFFu) {
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) {
                int v = accv[j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[j] += v;
            }
            uint32_t vk = accKv, vt = accTv;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { vk += __shfl_xor(vk, off, 64); vt += __shfl_xor(vt, off, 64); }
            accK += vk; accT += vt;
            if (accK != 0u) {
                const NSK_SCALAR uint32_t *pp = (const NSK_SCALAR uint32_t *)(g.tile_hdr + cur_prog);
#pragma unroll
                for (int j = 0; j < 4 * NCH; j++) {
                    const uint32_t s = pp[j];
                    const bool closes = (s >> 28) & 1u, fixed = (s >> 30) & 1u;      // uniform
                    if (closes && !fixed && lane == 0) {
                        const uint32_t code = (s >> 24) & 7u;
                        const long long span = code == 0u ? 0 : (code == 1u ? 1 : 2);            // hi - lo
                        const long long dG = (span * (long long)acc[j]) * g.grad_mul;            // Q31.32
                        const int wid = (int)(s & 0xFFFFFFu);
                        sink_add(sk.local, (unsigned long long *)&sk.G[wid], (unsigned long long)(dG + (sk.packed ? (long long)accK : 0)));
                        if (!sk.packed) sink_add(sk.local, &sk.K[wid], accK);
                        if (accT) sink_add(sk.local, &sk.T[wid], accT);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4 * NCH; j++) { acc[j] = 0; accv[j] = 0; }
        accK = 0u; accT = 0u; accKv = 0u; accTv = 0u;
    };
    if (bid0 < nfront) {                                                // (block-uniform) one quad that is not wide: this wave's tile of it
        const int Qr = (int)(rest.q[bid0] & 0x7FFFFFFFu);
        const SegEntry en = tab.e[seg_of_tile(tab, 4 * Qr)];
        cur_prog = en.prog;
        learn_tab_tile<NCH>(g, lp, en, 4 * Qr - en.tile_start - (int)(en.ntiles_lead >> 30) + wv, lane, acc, accK, accT);
        flush();
        close_sink<SMALLW>(g, sk);
        return;
    }
    for (int Q = q0 + wx; Q < q1;) {
        // the segment of the quad, and the wave's quads inside it (every segment is whole quads of the launch)
        const int sidx = seg_of_tile(tab, 4 * Q);
        const SegEntry en = tab.e[sidx];
        const int c_hi = sidx + 1 < NSK_SEG_MAX ? tab.e[sidx + 1].tile_start : tab.ntiles;
        const int Qe = min(q1, c_hi >> 2);
        const int lead = (int)(en.ntiles_lead >> 30), nt = (int)(en.ntiles_lead & 0x3FFFFFFFu);
        const int qs = en.tile_start >> 2;
        const int qin_lo = qs + (lead ? 1 : 0);
        const uint32_t qin_n = (uint32_t)(qs + ((lead + nt) >> 2) - qin_lo);
        const bool hasw = en.wide_off != NSK_NO_STREAM;
        const uint32_t zmask = en.zmask_ev & 0xFFu;
        const int ev = (int)(int8_t)(en.zmask_ev >> 8);
        const NSK_SCALAR uint32_t *wq = (const NSK_SCALAR uint32_t *)(g.seg_wide + (hasw ? en.wide_off : 0u)) + (hasw ? (size_t)(Q - qs) * ST : 0);
        const size_t wstep = hasw ? (size_t)wpx * ST : 0;
        int p0 = en.pos0 + (4 * Q - en.tile_start - lead) * 64;
        if (en.prog != cur_prog) { flush(); cur_prog = en.prog; }             // uniform, rare
        uint32_t cur[ST];
#pragma unroll
        for (int j = 0; j < ST; j++) cur[j] = wq[j];
        // the wave's copy of the segment's table entries (threshold: 27 + 26 bits, satisfied bits of both candidates)
#pragma unroll
        for (int r = 0; r < ZR; r++) {
            const uint32_t e = (uint32_t)lane + 64u * (uint32_t)r;
            if (e < (uint32_t)ZN) zt[e] = e <= zmask ? g.ztab[en.zoff + e] : uint4{0u, 0u, 0u, 0u};
        }
        asm volatile("" ::: "memory");                                  // (LDS executes a wave's accesses in order: no barrier)
        for (; Q < Qe; Q += wpx, p0 += 256 * wpx) {
            const bool flagged = hasw && cur[0] != 0xFFFFFFFFu, wide = flagged && (uint32_t)(Q - qin_lo) < qin_n;
            const uint32_t exc0 = cur[4 * NCH], nexc = cur[4 * NCH + 1], smask = cur[4 * NCH + 2];
            asm volatile("" :: "s"(cur[4 * NCH + 3]));
            WideLearnTrip<NCH> tr;
            const uint32_t l4 = 4u * (uint32_t)lane;
            if (wide) {
#pragma unroll
                for (int j = 0; j < 4 * NCH; j++) {
                    tr.xf[j] = *(const nsk_u32_una *)((const char *)g.val + (cur[j] + l4));
                    tr.xe[j] = *(const nsk_u32_una *)((const char *)g.val_evid + (cur[j] + l4));
                }
                tr.init = *(const uint32_t *)((const char *)g.p_init + ((uint32_t)p0 + l4));
            }
            if (Q + wpx < Qe) wq += wstep;                              // the next trip's descriptor, a scalar round trip ahead
#pragma unroll
            for (int j = 0; j < ST; j++) cur[j] = wq[j];
            if (!wide) {                                                // tile by tile -- unless a workgroup in front has the quad
                if (nfront) continue;
#pragma unroll 1
                for (int t = 0; t < 4; t++) learn_tab_tile<NCH>(g, lp, en, 4 * Q - en.tile_start - lead + t, lane, acc, accK, accT);
                continue;
            }
            // the four blocks of the lane's positions while the loads are in flight
            u32x4 rr[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) rr[i] = philox4x32(lp.k0, lp.k1, (uint32_t)p0 + l4 + (uint32_t)i, 0u, lp.s0, lp.s1);
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("" :: "v"(rr[i].x), "v"(rr[i].y), "v"(rr[i].z), "v"(rr[i].w));
            __builtin_amdgcn_sched_barrier(0);
            uint32_t idf4 = 0, ide4 = 0;
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++) { idf4 |= (tr.xf[j] & 0x01010101u) << j; ide4 |= (tr.xe[j] & 0x01010101u) << j; }
            idf4 &= smask * 0x01010101u;
            ide4 &= smask * 0x01010101u;
            for (uint32_t e = 0; e < nexc; e++) {                       // scalar loop, rare: the odd cells of the quad
                const NSK_SCALAR uint32_t *xp = (const NSK_SCALAR uint32_t *)g.wide_exc + 2 * (size_t)(exc0 + e);
                const uint32_t ex = xp[0], eid = xp[1];
                const uint32_t o = ex & 0xFFu, sh = 8u * (o & 3u) + ((ex >> 8) & 7u);
                if ((uint32_t)lane == (o >> 2)) {
                    idf4 = (idf4 & ~(1u << sh)) | (((uint32_t)(uint8_t)g.val[eid] & 1u) << sh);
                    ide4 = (ide4 & ~(1u << sh)) | (((uint32_t)(uint8_t)g.val_evid[eid] & 1u) << sh);
                }
            }
            uint32_t outf = 0, oute = 0, satf4 = 0, sate4 = 0, part4 = 0, trunc4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 ef = zt[(idf4 >> (8 * i)) & 0xFFu], ee = zt[(ide4 >> (8 * i)) & 0xFFu];
                const int init = (int)(int8_t)(tr.init >> (8 * i));                       // -1: padding position
                int evidence = init & 1;                                                  // learning.py:61-62
                if (ev != 1) evidence = k53(rr[i].z, rr[i].w) > ztab_K(ee) ? 1 : 0;       // 54-58
                const int proposal = k53(rr[i].x, rr[i].y) > ztab_K(ef) ? 1 : 0;          // 66-70
                const bool part = init >= 0 && (lp.learn_non_evidence || ev == 1);        // 71-72
                outf |= (uint32_t)proposal << (8 * i);
                oute |= (uint32_t)evidence << (8 * i);
                satf4 |= (part ? ((proposal ? (ef.z >> 8) : ef.z) & 0xFFu) : 0u) << (8 * i);
                sate4 |= (part ? ((evidence ? (ee.z >> 8) : ee.z) & 0xFFu) : 0u) << (8 * i);
                part4 |= (part ? 1u : 0u) << (8 * i);
            }
            if (lp.regularization == 1) {                                                 // 90
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const u32x4 tt = philox4x32(lp.k0, lp.k1, (uint32_t)p0 + l4 + (uint32_t)i, 1u, lp.s0, lp.s1);
                    trunc4 |= (((part4 >> (8 * i)) & 1u) && u53(tt.x, tt.y) < lp.inv_trunc ? 1u : 0u) << (8 * i);
                }
            }
            // (a padding position keeps what it holds: the stores cover whole dwords, so its bytes are written back)
            *(uint32_t *)((char *)g.val + ((uint32_t)p0 + l4)) = outf;
            *(uint32_t *)((char *)g.val_evid + ((uint32_t)p0 + l4)) = oute;
            accKv += (uint32_t)__popc(part4);
            accTv += (uint32_t)__popc(trunc4);
#pragma unroll
            for (int j = 0; j < 4 * NCH; j++)
                accv[j] += __popc(satf4 & (0x01010101u << j)) - __popc(sate4 & (0x01010101u << j));
        }
    }
    flush();
    close_sink<SMALLW>(g, sk);
}

template <typename VT, bool SMALLW>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_fast(DevGraph<VT> g, int pbegin, int pend,
                                                          int wb_base, const uint32_t *list, int ntiles,
                                                          LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6)));
    const int nwaves = (int)(gridDim.x * (NSK_BLOCK / 64));
    const int per = (ntiles + nwaves - 1) / nwaves;
    const int t1 = min(ntiles, (wave0 + 1) * per);
    for (int i = wave0 * per; i < t1; i++)
        learn_rest_tile<VT>(g, sk, pbegin, pend, wb_base, (int)__builtin_amdgcn_readfirstlane(list[i]), lp);
    (void)lane;
    close_sink<SMALLW>(g, sk);
}

// Learning over the general tiles [tile0, tile0 + ntiles) of a colour class
template <typename VT, bool SMALLW, int MAXC>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_general(DevGraph<VT> g, int pbegin, int pend,
                                                             int wb_base, int tile0, int ntiles,
                                                             int hb, int he, int hblocks,
                                                             const uint32_t *rest_list, int nrest, LearnParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) uint8_t lut[2048];
    __shared__ uint16_t gfacts[NSK_BLOCK / 64][NSK_LEARN_FACTS * 64];     // learn_tile_general: one array per wave
    load_gen_lut(lut);
    const GradSink sk = open_sink<SMALLW>(g, smem);
    const int lane = (int)(threadIdx.x & 63);
    if ((int)blockIdx.x < hblocks) {                      // hub blocks: one wave per hub position, strided
        const int hw0 = (int)(blockIdx.x * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
        for (int hp = hb + hw0; hp < he; hp += hblocks * (NSK_BLOCK / 64)) learn_hub<VT>(g, sk, lut, hp, lp.hub0 + (hp - hb), lp);
        close_sink<SMALLW>(g, sk);
        return;
    }
    const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x - hblocks) * (NSK_BLOCK / 64) + (threadIdx.x >> 6)));
    const int nwaves = (int)((gridDim.x - hblocks) * (NSK_BLOCK / 64));
    // XCD x (= blockIdx & 7: the hardware deals blocks to XCDs round-robin) walks the x-th eighth of the
    // tiles, so the member values a tile reads -- mostly positions near its own in every colour's
    // range -- stay in one XCD's L2
    const int xcd = (int)(blockIdx.x & 7);
    const int first = hblocks + ((xcd - (hblocks & 7) + 8) & 7);          // first tile block on this XCD
    const int nbx = first < (int)gridDim.x ? ((int)gridDim.x - 1 - first) / 8 + 1 : 0;
    const int wx = __builtin_amdgcn_readfirstlane((((int)blockIdx.x - first) >> 3) * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
    const int per8 = (ntiles + 7) >> 3;
    const int tend = min(ntiles, (xcd + 1) * per8);
    for (int t = xcd * per8 + wx; t < tend; t += nbx * (NSK_BLOCK / 64)) {
        const int tile = tile0 + t;
        const NSK_SCALAR uint32_t *tdp = (const NSK_SCALAR uint32_t *)(g.tiles + (wb_base + tile));
        const uint32_t tdx = tdp[0], tdz = tdp[2], tdw = tdp[3];
        const int p = pbegin + tile * 64 + lane;
        const bool valid = p < pend && g.p_vid[p] >= 0;
        learn_tile_general<VT, MAXC>(g, sk, lut, g.adj + tdx + lane, tdw, tdz, p, valid, lp, gfacts[threadIdx.x >> 6]);
    }
    // then the colour's uniform / shape tiles outside segment launches (a contiguous run per wave)
    const int per = (nrest + nwaves - 1) / nwaves;
    const int r1 = min(nrest, (wave0 + 1) * per);
    for (int i = wave0 * per; i < r1; i++)
        learn_rest_tile<VT>(g, sk, pbegin, pend, wb_base, (int)__builtin_amdgcn_readfirstlane(rest_list[i]), lp);
    close_sink<SMALLW>(g, sk);
}


// A long-list hub in the learning sweep, by a whole workgroup (block_hub_potentials): both chains'
// potentials with the ordered sums in wave 0, the draws, then the gradient pass over the entries
// spread over the four waves (as learn_heavy_variable_ep does with one).
template <typename VT>
__device__ __forceinline__ void block_hub_learn(const DevGraph<VT> &g, const GradSink &sk, const uint8_t *lut, int p,
                                                const uint4 hd, double *ws, uint16_t *fs, uint16_t *sel,
                                                const LearnParams &lp) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    if (g.p_vid[p] < 0) return;                                                    // (block-uniform)
    const uint32_t info = g.p_info[p];
    const int ev = NSK_INFO_EV(info), card = NSK_INFO_CARD(info);
    const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
    double pce = 0.0;
    if (ev != 1) pce = block_hub_potentials(g, lut, hd, g.val_evid, ws, fs);         // 54-58
    const double pcf = block_hub_potentials(g, lut, hd, g.val, ws, fs);              // 66-70
    if (wave == 0) {
        const int evidence = ev != 1 ? hub_draw(pce, card, u53(r.z, r.w)) : (int)g.p_init[p];   // 61-62
        const int proposal = hub_draw(pcf, card, u53(r.x, r.y));
        if (lane == 0) {
            g.val_evid[p] = (VT)evidence; g.val[p] = (VT)proposal;
            sel[0] = (uint16_t)evidence; sel[1] = (uint16_t)proposal;
        }
    }
    __syncthreads();
    if (!(lp.learn_non_evidence || ev == 1)) return;                               // 71-72
    const int evidence = (int)(int16_t)sel[0], proposal = (int)(int16_t)sel[1];
    bool truncate = false;
    if (lp.regularization == 1) {                                                  // 90
        const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
        truncate = u53(t.x, t.y) < lp.inv_trunc;
    }
    const int n = (int)hd.y, M = (int)(hd.z & 0xFFu), rows = 2 + M;
    const uint32_t *base = g.hub_adj + hd.x;
    for (int rr = wave; rr * 64 < n; rr += NSK_BLOCK / 64) {
        HubEntry ef, ee;
        hub_entry(g, lut, base, rows, rr, M, g.val, false, ef);
        hub_entry(g, lut, base, rows, rr, M, g.val_evid, false, ee);
        const bool mine = entry_visited(ef.d1, evidence, proposal);
        const long long diff = (long long)(proposal == ef.cstar ? ef.A : ef.B) -
                               (long long)(evidence == ee.cstar ? ee.A : ee.B);
        const bool have = rr * 64 + lane < n && mine && !g.w_fixed[ef.wid];                        // 100-101
        accumulate_gradient(sk, have, (int)ef.wid, diff * g.grad_mul, truncate);
    }
}

// (Round 5 measured three variants of this launch, none kept -- tools/sessions/r5_s02.sh, r5_s03.sh: the gradient pass
// one lane per VARIABLE from facts / weight ids kept in the slots, no second read of the rows and no third barrier:
// 5M LR 114.5 -> 122.7 us per class, 50M 1125 -> 1121; LDS value windows and non-temporal weight gathers:
// nsk_compile.h ep_win, ep_pass.)
// Learning over the entry-parallel groups of a colour class (k_gibbs_ep's layout, passes and phases):
//   1. one lane per list entry, BOTH chains: (weight, owner, free-chain facts, evidence-chain facts)
//      into the LDS slot [position in the list][variable];
//   2. one lane per variable: free-chain and evidence-chain potentials in list order -> proposal and
//      evidence (learning.py:54-70), stores, and (evidence, proposal, takes part, truncation coin)
//      into LDS;
//   3. one lane per list entry again: the entry's gradient value(proposal | free chain) -
//      value(evidence | evidence chain) from the saved facts, for the entries sample_and_sgd visits
//      (entry_visited), into the wave-aggregated accumulators.
// A group with more than 8 entries per variable takes two passes of (1, 2); phase 3 then runs over the
// second pass's entries (their facts are in LDS) and, after redoing phase 1 for them, over the first's.
// Dynamic LDS: the SMALLW accumulators only.
// Four waves per SIMD: with the wave's index a scalar (ep_pass) the eight-candidate instantiation needs 139 vector
// registers -- three waves --, and capped at 128 it spills 16 bytes: 5M LR graph 121.1 -> 113.3 us per class (5.16 ->
// 5.52e9 updates/s), 50M 1114 -> 1051 (5.61 -> 5.94e9); tools/sessions/r5_s29.sh.  (The two-candidate instantiation
// has 114 and is at four already; at 152 registers, before, the cap was measured slower.)
#ifndef NSK_EP_WPE_L
#define NSK_EP_WPE_L 4
#endif
#define NSK_EP_ATTR_L __attribute__((amdgpu_waves_per_eu(NSK_EP_WPE_L, NSK_EP_WPE_L)))
template <typename VT, bool SMALLW, int MAXC>
__device__ __forceinline__ void learn_ep_body(const DevGraph<VT> &g, int pbegin, int pend, int wb_base,
                                              int tile0, int ntiles, int ngroups, int group0, int gblocks,
                                              int hb, int he, int hblocks, int nbh, int bh0,
                                              const uint32_t *rest_list, int nrest, const LearnParams &lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) double ws[NSK_EP_LIST * 256];
    __shared__ __attribute__((aligned(16))) uint32_t fs[NSK_EP_LIST * 256];
    __shared__ uint16_t sel[256];
    __shared__ __attribute__((aligned(16))) uint8_t lut[2048];
    load_gen_lut(lut);
    const GradSink sk = open_sink<SMALLW>(g, smem);
    if ((int)blockIdx.x < nbh) {                          // one long-list hub per block
        const int hp = (int)__builtin_amdgcn_readfirstlane(g.bighub_pos[bh0 + (int)blockIdx.x]);
        const NSK_SCALAR uint32_t *hdp = (const NSK_SCALAR uint32_t *)(g.hub_desc + lp.hub0 + (hp - hb));
        const uint4 hd = {hdp[0], hdp[1], hdp[2], hdp[3]};
        block_hub_learn<VT>(g, sk, lut, hp, hd, ws, (uint16_t *)fs, sel, lp);
        close_sink<SMALLW>(g, sk);
        return;
    }
    if ((int)blockIdx.x < hblocks) {                      // hub blocks: one wave per hub position, strided
        const int hw0 = (int)((blockIdx.x - nbh) * (NSK_BLOCK / 64) + (threadIdx.x >> 6));
        for (int hp = hb + hw0; hp < he; hp += (hblocks - nbh) * (NSK_BLOCK / 64)) learn_hub<VT>(g, sk, lut, hp, lp.hub0 + (hp - hb), lp);
        close_sink<SMALLW>(g, sk);
        return;
    }
    const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x - hblocks) * (NSK_BLOCK / 64) + (threadIdx.x >> 6)));
    const int nwaves = gblocks * (NSK_BLOCK / 64);
    // phase 1 of one pass: both chains' facts and the weight into the slots
    auto entries = [&](uint32_t sub, uint32_t rowsw) {
        for (int i = (int)threadIdx.x; i < NSK_EP_LIST * 256; i += NSK_BLOCK) fs[i] = 14u;      // owned by no candidate
        __syncthreads();
        ep_pass<VT, true, 1, false, true, 2>(g, g.val, g.val_evid, sub, rowsw, nullptr, nullptr, nullptr, false,
            [&](uint32_t w0, uint32_t d1, const GenChain &a, const GenChain &b, double w) {
                int cstar, A, B;
                a.close(d1, lut, cstar, A, B);
                uint32_t fx = ep_facts(cstar, A, B) << 4;
                b.close(d1, lut, cstar, A, B);
                fx |= ep_facts(cstar, A, B) << 12;
                const uint32_t ks = (d1 >> 14) & 15u;
                if (ks != 14u) {
                    const uint32_t slot = NSK_EP_SLOT(w0, d1);
                    ws[slot] = w;
                    fs[slot] = ks | fx;
                }
            });
        __syncthreads();
    };
    // phase 3 of one pass: gradients of its entries from the facts in the slots
    auto gradients = [&](uint32_t sub, uint32_t rowsw) {
        ep_pass<VT, false, 0, false, false, 2>(g, g.val, g.val, sub, rowsw, nullptr, nullptr, nullptr, false,
            [&](uint32_t w0, uint32_t d1, const GenChain &, const GenChain &, double) {
                const uint32_t sv = sel[(d1 >> 23) & 255u];
                const int evidence = (int)(sv & 15u), proposal = (int)((sv >> 4) & 15u);
                const uint32_t f = fs[NSK_EP_SLOT(w0, d1)];
                const int cf = (int)((f >> 4) & 15u), Af = (int)((f >> 8) & 3u) - 1, Bf = (int)((f >> 10) & 3u) - 1;
                const int ce = (int)((f >> 12) & 15u), Ae = (int)((f >> 16) & 3u) - 1, Be = (int)((f >> 18) & 3u) - 1;
                const long long diff = (long long)(proposal == cf ? Af : Bf) - (long long)(evidence == ce ? Ae : Be);
                // (a dataType-0 entry's visit is counted structurally: with a zero gradient -- the common
                // case -- it has nothing to add)
                const bool counted = lp.kstat && ((d1 >> 14) & 15u) == 15u;
                const bool have = (sv & 256u) && ((d1 >> 14) & 15u) != 14u && entry_visited(d1, evidence, proposal) &&
                                  !(d1 >> 31) && !(counted && diff == 0);                    // 100-101
                // (the entry's weight is still in its LDS slot from phase 1: a weight updated in place needs no reload)
                accumulate_gradient(sk, have, (int)NSK_EP_WID(w0), diff * g.grad_mul, (sv & 512u) != 0u, !counted,
                                    sk.w_direct != nullptr, sk.w_direct ? ws[NSK_EP_SLOT(w0, d1)] : 0.0);
            });
    };
    if ((int)blockIdx.x < hblocks + gblocks) {
        const EpWalk wk = ep_walk(ngroups, hblocks, gblocks);
        for (int li = wk.li; li < wk.lend; li += wk.step) {
            const int gi = ep_group(wk, li, ngroups);
            if (gi < 0) continue;
            const NSK_SCALAR uint32_t *gdp = (const NSK_SCALAR uint32_t *)(g.ep_desc + group0 + gi);
            const uint32_t gsub = gdp[0], grows0 = gdp[1], gmax = gdp[2], grows1 = gdp[3];
            const int ne = (int)(gmax & 255u);
            const int tile = tile0 + 4 * gi + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (a scalar: ep_pass)
            const bool tile_ok = tile < tile0 + ntiles;                           // wave-uniform
            const int p = pbegin + tile * 64 + (int)(threadIdx.x & 63);
            const bool valid = tile_ok && p < pend && g.p_vid[p] >= 0;
            const uint32_t info = valid ? g.p_info[p] : (2u << 9);
            const uint32_t tdw = tile_ok ? *(const NSK_SCALAR uint32_t *)((const uint32_t *)(g.tiles + (wb_base + tile)) + 3) : 0u;
            const int maxcard = (int)((tdw >> 12) & 15u);
            const int ev = NSK_INFO_EV(info), card = NSK_INFO_CARD(info);
            const bool need_evid = __ballot(valid && ev != 1) != 0;               // wave-uniform
            GenPot<MAXC> pf, pe;
            pf.clear(); pe.clear();
            uint32_t sub = gsub;
            const bool two = ne > NSK_EP_LIST;
            for (int pass = 0; pass * NSK_EP_LIST < ne; pass++) {
                const uint32_t rowsw = pass ? grows1 : grows0;
                __syncthreads();                           // (the previous group / pass is done with the slots)
                entries(sub, rowsw);
                const int nacc = min(NSK_EP_LIST, ne - pass * NSK_EP_LIST);
                if (tile_ok)
                    for (int o = 0; o < nacc; o++) {
                        const uint32_t f = fs[o * 256 + (int)threadIdx.x];
                        const double w = ws[o * 256 + (int)threadIdx.x];
                        pf.add_ks(maxcard, (int)(f & 15u), w, (int)((f >> 4) & 15u), (int)((f >> 8) & 3u) - 1, (int)((f >> 10) & 3u) - 1);
                        if (need_evid)
                            pe.add_ks(maxcard, (int)(f & 15u), w, (int)((f >> 12) & 15u), (int)((f >> 16) & 3u) - 1, (int)((f >> 18) & 3u) - 1);
                    }
                if (pass == 0 && two) sub += (uint32_t)ep_pass_subrows(rowsw);
            }
            uint32_t mysel = 0;
            if (tile_ok) {
                const u32x4 r = philox4x32(lp.k0, lp.k1, (uint32_t)p, 0u, lp.s0, lp.s1);
                const int proposal = pf.draw(maxcard, card, u53(r.x, r.y));                    // 66-70
                int evidence = valid ? (int)g.p_init[p] : 0;                                   // 61-62
                if (need_evid && ev != 1) evidence = pe.draw(maxcard, card, u53(r.z, r.w));    // 54-58
                if (valid) {
                    g.val_evid[p] = (VT)evidence;
                    g.val[p] = (VT)proposal;
                }
                const bool part = valid && (lp.learn_non_evidence || ev == 1);                 // 71-72
                bool truncate = false;
                if (lp.regularization == 1) {                                                  // 90
                    const u32x4 t = philox4x32(lp.k0, lp.k1, (uint32_t)p, 1u, lp.s0, lp.s1);
                    truncate = part && (u53(t.x, t.y) < lp.inv_trunc);
                }
                mysel = ((uint32_t)evidence & 15u) | (((uint32_t)proposal & 15u) << 4) | (part ? 256u : 0u) | (truncate ? 512u : 0u);
            }
            sel[threadIdx.x] = (uint16_t)mysel;
            __syncthreads();
            gradients(sub, two ? grows1 : grows0);         // the last pass's facts are in the slots
            if (two) {                                     // the first pass's entries: their facts again, then their gradients
                __syncthreads();
                entries(gsub, grows0);
                gradients(gsub, grows0);
            }
        }
    }
    // then the colour's uniform / shape tiles outside segment launches (a contiguous run per wave of the
    // group and rest blocks)
    // XCD x (= blockIdx & 7: how the hardware deals workgroups; for speed only) takes the x-th eighth of the
    // list, a contiguous run per wave: the list is in position order and shape classes are formed per id
    // range (nsk_compile.cpp "shape_parts"), so the values and weights an XCD's tiles gather lie in a
    // correspondingly narrow stretch of every colour -- one L2's worth
    const int nb_all = (int)gridDim.x - hblocks;
    int r0, r1;
    if (nb_all >= 8) {
        const int xcd = (int)(blockIdx.x & 7);
        const int first = hblocks + ((xcd - (hblocks & 7) + 8) & 7);          // first such block on this XCD
        const int nwx = (((int)gridDim.x - 1 - first) / 8 + 1) * (NSK_BLOCK / 64);
        const int wx = __builtin_amdgcn_readfirstlane((((int)blockIdx.x - first) >> 3) * (NSK_BLOCK / 64) + (int)(threadIdx.x >> 6));
        const int per8 = (nrest + 7) >> 3;
        const int t0 = min(nrest, xcd * per8), tend = min(nrest, t0 + per8);
        const int per = (tend - t0 + nwx - 1) / nwx;
        r0 = min(tend, t0 + wx * per);
        r1 = min(tend, r0 + per);
    } else {
        const int nw_all = nb_all * (NSK_BLOCK / 64);
        const int per = (nrest + nw_all - 1) / nw_all;
        r0 = min(nrest, wave0 * per);
        r1 = min(nrest, r0 + per);
    }
    __syncthreads();                                       // (the last group's gradient pass is done with `ws`)
    for (int i = r0; i < r1; i++)
        learn_rest_tile<VT>(g, sk, pbegin, pend, wb_base, (int)__builtin_amdgcn_readfirstlane(rest_list[i]), lp, ws);
    (void)nwaves;
    close_sink<SMALLW>(g, sk);
}

#define NSK_LEP_PARAMS DevGraph<VT> g, int pbegin, int pend, int wb_base, int tile0, int ntiles, int ngroups, int group0, int gblocks, \
                       int hb, int he, int hblocks, int nbh, int bh0, const uint32_t *rest_list, int nrest, LearnParams lp
#define NSK_LEP_FORWARD g, pbegin, pend, wb_base, tile0, ntiles, ngroups, group0, gblocks, hb, he, hblocks, nbh, bh0, rest_list, nrest, lp
template <typename VT, bool SMALLW, int MAXC>
__global__ __launch_bounds__(NSK_BLOCK) void k_learn_ep(NSK_LEP_PARAMS) { learn_ep_body<VT, SMALLW, MAXC>(NSK_LEP_FORWARD); }
// the same at four waves per SIMD (the eight-candidate instantiation: see NSK_EP_WPE_L above)
template <typename VT, bool SMALLW, int MAXC>
__global__ __launch_bounds__(NSK_BLOCK) NSK_EP_ATTR_L void k_learn_ep_w4(NSK_LEP_PARAMS) { learn_ep_body<VT, SMALLW, MAXC>(NSK_LEP_FORWARD); }
#undef NSK_LEP_PARAMS
#undef NSK_LEP_FORWARD


static __global__ __launch_bounds__(NSK_BLOCK) void k_apply_weights(double *w, const double *w_in, long long *G, uint32_t *K,
                                                             uint32_t *T, int nweight, double step,
                                                             int regularization, double reg_param,
                                                             double truncation, int packed, double cap,
                                                             unsigned int *clipped, int copies, double grad_inv,
                                                             const uint32_t *kstat_ev, const uint32_t *kstat_other,
                                                             const int32_t *widx, int nidx) {
    // widx: the weights to walk (the ones that are not updated in place, nsk_compile.h multi_wids); null: all
    const int at_ = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (at_ >= (widx ? nidx : nweight)) return;
    const int i = widx ? widx[at_] : at_;
    long long gsum = 0;
    // structural visit counts of the class (nsk_compile.h ep_kstat): visits the kernels did not count
    unsigned long long k = (kstat_ev ? (unsigned long long)kstat_ev[i] : 0ull) +
                           (kstat_other ? (unsigned long long)kstat_other[i] : 0ull), t = 0;
    for (int x = 0; x < copies; x++) {                  // the XCDs' private copies (cleared as they are read)
        const size_t at = (size_t)x * nweight + i;
        long long gx = G[at];
        if (packed) {                   // visits in the low half, integer gradient sum in the high half
            const unsigned long long kx = (unsigned long long)gx & 0xFFFFFFFFull;
            if (gx) G[at] = 0;
            gx -= (long long)kx;
            k += kx;
        } else {
            const uint32_t kx = K[at];
            if (kx || gx) { G[at] = 0; K[at] = 0; }          // (a structurally counted visit adds to G only)
            k += kx;
        }
        gsum += gx;
        if (regularization == 1) {      // only L1 ever counts truncations
            const uint32_t tx = T[at];
            if (tx) T[at] = 0;
            t += tx;
        }
    }
    if (k == 0) {                       // untouched in this class
        if (w != w_in) w[i] = w_in[i];
        return;
    }
    w[i] = apply_update(w_in[i], gsum, (uint32_t)k, (uint32_t)t, step, regularization, reg_param, truncation, cap, clipped, grad_inv);
}

}  // namespace nsk
