// nsk_learn.hip -- learning sweep driver: replaces the epoch loop around run_pool(learnthread) at
// numbskull/factorgraph.py:194-206; kernels in nsk_kernels_learn.h.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "nsk_internal.h"
#include "nsk_kernels_learn.h"
#include "nsk_kernels_misc.h"

using namespace nsk;

#ifndef NSK_LEARN_TPW
#define NSK_LEARN_TPW 2          // tiles per trip of the learning table kernel (1: 41.6 us, 2: 41.6 us, 4: 46.7 us
                                 // per 10M-grid class before the kernel was pipelined)
#endif

// the fast-path refresh after a weight update of the large-table path, for one weight set on one stream
static void refresh_after_update(nsk_graph *g, int set, hipStream_t st) {
    const int n = (int)g->c.tile_hdr.size();
    // (n <= 8: the closing pad alone -- a graph of general tiles and entry-parallel groups has no slot programs, and
    // the launch would be 4.7 us per class of nothing: 4 % of a 5M LR graph's learning sweep)
    if (n > 8 && g->c.nfast > 0 && g->c.nweight > 0) {
        k_refresh_prog_weights<<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, st>>>(
            g->tile_hdr, set ? g->w1 : g->w, set ? g->prog_w1 : g->prog_w, n);
        nsk_refresh_ztab(g, set, st);
    }
}

// The chromatic learning sweep is instantiated four times (value type x accumulator flavour) and every
// instantiation carries a dozen kernels: the Makefile compiles this file once per instantiation
// (-DNSK_LEARN_PART=0..3, in parallel; part 0 also holds the entry point) -- one translation unit with all four
// (NSK_LEARN_PART undefined: the ablation builds) takes six minutes.
#ifndef NSK_LEARN_PART
#define NSK_LEARN_PART -1
#endif
#define NSK_LEARN_HAS(P) (NSK_LEARN_PART == -1 || NSK_LEARN_PART == (P))
#define NSK_LEARN_SIG nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization, \
                      double reg_param, int64_t truncation, int learn_non_evidence
template <typename VT, bool SMALLW>
int nsk_learn_chromatic(NSK_LEARN_SIG);
#if NSK_LEARN_PART != -1        // the instantiations of the other parts are theirs
#if !NSK_LEARN_HAS(0)
extern template int nsk_learn_chromatic<int8_t, false>(NSK_LEARN_SIG);
#endif
#if !NSK_LEARN_HAS(1)
extern template int nsk_learn_chromatic<int8_t, true>(NSK_LEARN_SIG);
#endif
#if !NSK_LEARN_HAS(2)
extern template int nsk_learn_chromatic<int32_t, false>(NSK_LEARN_SIG);
#endif
#if !NSK_LEARN_HAS(3)
extern template int nsk_learn_chromatic<int32_t, true>(NSK_LEARN_SIG);
#endif
#endif

template <typename VT, bool SMALLW>
int nsk_learn_chromatic(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                        double reg_param, int64_t truncation, int learn_non_evidence) {
    const size_t nphase = g->c.phase_start.size() - 1;
    const int nw = (int)g->c.nweight;
    const size_t shmem = SMALLW ? (size_t)nw * 16 : 0;
    LearnParams lp;
    lp.regularization = regularization;
    lp.learn_non_evidence = learn_non_evidence;
    lp.inv_trunc = 1.0 / (double)truncation;
    lp.k0 = (uint32_t)g->seed; lp.k1 = (uint32_t)(g->seed >> 32);
    lp.kstat = 0;
    g->adj_wt_skip = true;          // the learning kernels gather weights themselves
    nsk_refresh_prog_weights(g);
    // One-class lag (nsk_set_learn_lag): class c of this call samples from weight set c & 1 -- the weights
    // as of the end of class c - 2 -- and adds into accumulator set c & 1; its update U(c) reads the weights
    // of set (c + 1) & 1 (end of class c - 1) and writes set c & 1, and is enqueued BEHIND class c + 1,
    // which does not read it: fused into that class's table launch when there is one (block 0 of
    // k_learn_seg_tab, beside the sampling), else as a launch of its own after the class.  Both sets start
    // from the call's weights.  (A second stream with events around every class cost more than the update:
    // 34 against 28.5 us per 10M-grid class.)
    // (only handles that accumulate in LDS -- at most NSK_SMALLW weights: the grids -- run lagged: there the
    // update rides in the next class's launch; with a large weight table it is a launch of its own either
    // way, and the in-kernel updates of single-factor weights below want one weight set)
    const bool lag = g->learn_lag && nw > 0 && SMALLW;
    DevGraph<VT> dv[2];
    dv[0] = view<VT>(g);
    dv[1] = dv[0];
    if (lag) {
        int rc = nsk_ensure_lag_sets(g);
        if (rc) return rc;
        dv[1].w = g->w1; dv[1].prog_w = g->prog_w1; dv[1].ztab = g->ztab1;
        dv[1].G = g->G1; dv[1].K = g->K1; dv[1].T = g->T1;
        dv[1].part_G = g->part_G1; dv[1].part_K = g->part_K1; dv[1].part_T = g->part_T1;
        HIPCHECK(hipMemcpyAsync(g->w1, g->w, (size_t)nw * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        if (!g->c.tile_hdr.empty())
            HIPCHECK(hipMemcpyAsync(g->prog_w1, g->prog_w, 2 * g->c.tile_hdr.size() * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        if (g->c.nztab)
            HIPCHECK(hipMemcpyAsync(g->ztab1, g->ztab, (size_t)g->c.nztab * sizeof(uint4), hipMemcpyDeviceToDevice, g->stream));
    }
    int64_t cls = 0;                // classes launched in this call
    struct Pending {                // the update of the previous class, not yet enqueued
        bool valid = false;
        int set = 0;
        bool tabs_here = false, kstat = false;
        size_t ph = 0;
        double step = 0.0;
        ApplyArgs aa;
    } pend;
    auto launch_update = [&](const Pending &u) {      // as launches of their own on the main stream
        const DevGraph<VT> &du = dv[u.set];
        if (SMALLW) {
            k_apply_bins<<<dim3(1), dim3(NSK_BLOCK), 0, g->stream>>>(u.aa);
            if (g->c.nfast > 0 && !u.tabs_here) nsk_refresh_ztab(g, u.set, g->stream);   // big tables: own launch
        } else {
            // (with direct weights only the others are walked -- none at all when every weight has one factor)
            const bool direct = g->c.ndirect > 0;
            const int nwalk = direct ? (int)g->c.multi_wids.size() : nw;
            if (nwalk > 0) {
                k_apply_weights<<<dim3((nwalk + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                    du.w, u.aa.w_in, du.G, du.K, du.T, nw, u.step, regularization, reg_param, (double)truncation,
                    (!SMALLW && g->c.packed_grad) ? 1 : 0, g->learn_cap, g->clip_count, g->acc_copies, du.grad_inv,
                    u.kstat ? du.ep_kstat + (size_t)(2 * u.ph) * (size_t)nw : nullptr,
                    (u.kstat && learn_non_evidence) ? du.ep_kstat + (size_t)(2 * u.ph + 1) * (size_t)nw : nullptr,
                    direct ? g->multi_wids : nullptr, nwalk);
                refresh_after_update(g, u.set, g->stream);
            }
        }
    };
    ApplyArgs no_update;
    memset(&no_update, 0, sizeof(no_update));
    for (int64_t s = 0; s < nsweeps; s++) {
        lp.s0 = (uint32_t)g->sweep; lp.s1 = nsk_sweep_hi(g);
        for (size_t ph = 0; ph < nphase; ph++) {
            const int fb = (int)g->c.phase_start[ph], fe = (int)g->c.phase_fast_end[ph];
            const int e = (int)g->c.phase_end[ph];
            if (e <= fb) continue;
            const int set = lag ? (int)(cls & 1) : 0;
            DevGraph<VT> d = dv[set];
            // the parameters of this class's update, for the weights the kernels update in place (w_direct)
            d.upd_step = step; d.upd_regularization = regularization; d.upd_reg_param = reg_param;
            d.upd_truncation = (double)truncation; d.upd_cap = g->learn_cap;
            d.upd_a1 = 1.0 / (1.0 + reg_param * step);
            lp.hub0 = (int)g->c.phase_hub_base[ph];
            const int ntiles = (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]);
            const int ndyn = (int)(g->c.phase_dyn_base[ph + 1] - g->c.phase_dyn_base[ph]);
            const int he = (int)g->c.phase_heavy_end[ph];
            ColourStreams cs(g, !g->no_overlap);
            // the weight update of the class (SMALLW): arguments of k_apply_bins
            const bool tabs_here = g->c.nfast > 0 && g->c.nztab <= 2048;
            ApplyArgs aa;
            memset(&aa, 0, sizeof(aa));
            aa.w = d.w; aa.w_in = lag ? dv[set ^ 1].w : d.w;
            aa.part_G = d.part_G; aa.part_K = d.part_K; aa.part_T = d.part_T;
            aa.nweight = nw; aa.step = step; aa.regularization = regularization; aa.reg_param = reg_param;
            aa.truncation = (double)truncation; aa.prog = g->tile_hdr; aa.prog_w = const_cast<double *>(d.prog_w);
            aa.nprog = g->c.nfast > 0 ? (int)g->c.tile_hdr.size() : 0; aa.zp = g->zprogs;
            aa.nzp = tabs_here ? (int)g->c.zprogs.size() : 0; aa.nztab = tabs_here ? (int)g->c.nztab : 0;
            aa.ztab = const_cast<uint4 *>(d.ztab); aa.cap = g->learn_cap; aa.clipped = g->clip_count; aa.grad_inv = d.grad_inv;
            if (e > he) {               // variables outside the fast path: generic kernel, range mode
                const int nitems = (e - he + 63) / 64;
                const int grid = std::min(NSK_LEARN_GEN_BLOCKS, (nitems + 3) / 4);
                k_learn_phase<VT, SMALLW, true><<<dim3(grid), dim3(NSK_BLOCK), shmem, cs.side(1)>>>(
                    d, he, e, nullptr, nitems, lp);
                g->launches++;
            }
            const int gt0 = (int)g->c.phase_gen_tile[ph];
            const int gtb = (int)g->c.phase_gen_bin_tile[ph];
            // hubs ride as extra blocks of a general-tile launch when the class has one
            const bool hubs_in_general = (gtb > gt0 || ntiles > gtb) && he > fe;
            const int hbl = hubs_in_general ? std::min(NSK_LEARN_HEAVY_BLOCKS, (he - fe + 3) / 4) : 0;
            if (he > fe && !hubs_in_general) {   // hubs: one wave per variable
                const int grid = std::min(NSK_LEARN_HEAVY_BLOCKS, (he - fe + 3) / 4);
                k_learn_heavy<VT, SMALLW><<<dim3(grid), dim3(NSK_BLOCK), shmem, cs.side(2)>>>(d, fe, he, lp);
                g->launches++;
            }
            // the colour's uniform / shape tiles outside segment launches ride in the general launch
            const int nlrest = (int)(g->c.phase_learn_rest_base[ph + 1] - g->c.phase_learn_rest_base[ph]);
            const bool rest_in_general = ntiles > gt0 && nlrest > 0;
            const uint32_t *lrest = g->learn_rest_tiles + g->c.phase_learn_rest_base[ph];
            // a class with a lot of both general tiles and other tiles runs the two groups side by side
            // (side stream 0); smaller ones are not worth the fork / join events
            int other_tiles = rest_in_general ? 0 : nlrest;
            for (const Compiled::SegLaunch &sl : g->c.learn_seg) if (sl.phase == (int)ph) other_tiles += sl.tile_start[sl.n];
            const bool general_aside = ntiles - gt0 >= 2048 && other_tiles >= 2048 && !g->no_overlap;
            // as in inference: a class with categorical tiles walks all its general tiles in one
            // launch of the 8-candidate kernel on the main stream (+10 % over two concurrent launches)
            const bool one_lg = gtb > gt0;
            const bool ep = fe > fb && g->c.phase_ep[ph] && ntiles > gt0;
            // structural visit counts: the entry-parallel launch skips zero-gradient visits of dataType-0
            // variables and the weight update adds their counts (not with L1: its truncation coins are
            // counted per visit)
            const bool kstat_here = ep && !SMALLW && d.ep_kstat != nullptr && regularization != 1;
            lp.kstat = kstat_here ? 1 : 0;
            if (ep) {                   // entry-parallel groups: hubs, the general tiles and the rest tiles
                const int ngroups = (int)(g->c.phase_ep_base[ph + 1] - g->c.phase_ep_base[ph]);
                // grid: 10 workgroups per CU (4 are resident -- 128 with the cap of k_learn_ep_w4 / 114 vector registers; 153 / 111 and 3 or 4 until round 5 --, the others
                // start as those end: dynamic dealing of uneven groups).  Per class (NSK_EP_PER_CU), 5M LR graph:
                // 3 workgroups per CU 142.4 us, 4 149.2, 5 139.5, 6 135.4, 7 134.2, 8 132.9, 10 134.7, 16 135.6;
                // 50M LR graph: 4 1362 us, 7 1247, 10 1189
                const char *pcu_env = nsk::diag_env("NSK_EP_PER_CU");            // (diagnostic: workgroups per CU)
                // -- and at 5M, twice each (tools/sessions/history/r4_s26.sh): 10 per CU 4.95 / 4.97e9 updates/s, 7 per CU
                // 5.04 / 5.06e9.  Hence 10 when a workgroup walks four groups or more, 7 otherwise (the shards
                // of an 8-rank run of the 50M graph are of the second kind).
                const int per_cu = pcu_env ? std::max(1, std::min(32, atoi(pcu_env))) : (ngroups >= 4 * 2560 ? 10 : 7);
                const int gblocks = 8 * ((std::min(256 * per_cu, ngroups) + 7) / 8);
                const int nbh = (int)(g->c.phase_bighub_base[ph + 1] - g->c.phase_bighub_base[ph]);   // a block per long-list hub
                const int hbl_ep = nbh + hbl;
                // the colour's rest tiles (shape tiles: graphs with individual weights) are walked by all of these
                // workgroups, one wave per tile: when the groups alone bring too few, more workgroups (they skip the
                // group walk) -- the weighted boolean graph has 750 groups and 7 000 rest tiles per colour
                const int nrest_here = rest_in_general ? nlrest : 0;
                const int rblocks = std::max(0, 8 * ((std::min(2560, (nrest_here + 3) / 4) + 7) / 8) - gblocks);
                const int grid = gblocks + rblocks + hbl_ep;
                hipStream_t st = general_aside ? cs.side(0) : g->stream;
#define NSK_LEP(KERNEL, MAXC) KERNEL<VT, SMALLW, MAXC><<<dim3(grid), dim3(NSK_BLOCK), shmem, st>>>( \
                    d, fb, fe, (int)g->c.phase_wb_base[ph], gt0, ntiles - gt0, ngroups, (int)g->c.phase_ep_base[ph], gblocks, \
                    fe, he, hbl_ep, nbh, (int)g->c.phase_bighub_base[ph], lrest, rest_in_general ? nlrest : 0, lp)
                // (a colour's all-binary groups -- its tail: categorical lanes come first -- in a launch of their own with the
                // two-candidate kernel, 114 vector registers and four waves per SIMD instead of 152 and three: 50M LR graph
                // 5.57 -> 5.73e9 updates/s, 5M 5.13 -> 4.17e9 with the split forced (two tails per class); inference, 86 / 5
                // against 102 / 4: 1.540 / 1.540e10 -- tools/sessions/r5_s24.sh; not kept)
                if (one_lg) NSK_LEP(k_learn_ep_w4, 8); else NSK_LEP(k_learn_ep, 2);
#undef NSK_LEP
                g->launches++;
            }
            if (gtb > gt0 && !ep) {            // general tiles with categorical lanes
                const int nt8 = one_lg ? ntiles - gt0 : gtb - gt0;
                const int grid = 8 * ((std::min(NSK_LEARN_GENERAL_BLOCKS / 2, (nt8 + 3) / 4) + 7) / 8) + hbl;   // whole rounds of XCDs
                k_learn_general<VT, SMALLW, 8><<<dim3(grid), dim3(NSK_BLOCK), shmem, (one_lg && !general_aside) ? g->stream : cs.side(0)>>>(
                    d, fb, fe, (int)g->c.phase_wb_base[ph], gt0, nt8, fe, he, hbl, lrest,
                    (rest_in_general && one_lg) ? nlrest : 0, lp);
                g->launches++;
            }
            if (ntiles > gtb && !one_lg && !ep) {   // all-binary general tiles
                const int hb2 = gtb > gt0 ? 0 : hbl;        // no categorical launch: the hubs come here
                const int grid = 8 * ((std::min(NSK_LEARN_GENERAL_BLOCKS / 2, (ntiles - gtb + 3) / 4) + 7) / 8) + hb2;
                k_learn_general<VT, SMALLW, 2><<<dim3(grid), dim3(NSK_BLOCK), shmem, general_aside ? cs.side(0) : g->stream>>>(
                    d, fb, fe, (int)g->c.phase_wb_base[ph], gtb, ntiles - gtb, fe, he, hb2, lrest,
                    (rest_in_general && !(one_lg)) ? nlrest : 0, lp);
                g->launches++;
            }
            for (const Compiled::SegLaunch &sl : g->c.learn_seg) {       // homogeneous segments
                if (sl.phase != (int)ph) continue;
                const bool use_tab = sl.tab && g->values_regular;
                SegTable tab;
                memset(&tab, 0, sizeof(tab));
                tab.n = sl.n;
                // (mostly) wide quads: the wide learning kernel, its tiles numbered in whole quads like the inference
                // launches (lead = dead tiles in front of a segment that does not start on a quad boundary)
                if constexpr (sizeof(VT) == 1) if (use_tab) {
                    // (prepared once per handle: walking the launch's quad descriptors on the host for every launch of every sweep
                    //  cost more than the launch itself from a few million variables on)
                    if (g->learn_wide_plans.size() != g->c.learn_seg.size()) g->learn_wide_plans.assign(g->c.learn_seg.size(), NskLearnWidePlan());
                    NskLearnWidePlan &wp = g->learn_wide_plans[(size_t)(&sl - g->c.learn_seg.data())];
                    if (wp.key != 1) {
                        wp = NskLearnWidePlan();
                        wp.key = 1;
                        SegTable &wt = wp.tab;
                        memset(&wt, 0, sizeof(wt));
                        wt.n = sl.n;
                        int vt = 0, nwide = 0;
                        for (int i = 0; i < NSK_SEG_MAX; i++) {
                            SegEntry &en = wt.e[i];
                            en.tile_start = vt;
                            if (i >= sl.n) continue;
                            const int nt_i = sl.tile_start[i + 1] - sl.tile_start[i];
                            const int64_t pos0 = sl.pos0[i];
                            const int lead = (int)((pos0 / 64) & 3);
                            en.ntiles_lead = (uint32_t)nt_i | ((uint32_t)lead << 30);
                            en.pos0 = sl.pos0[i]; en.adj_off = sl.adj_off[i]; en.prog = sl.prog[i]; en.zoff = sl.zoff[i];
                            en.zmask_ev = (sl.zmask[i] & 0xFFu) | (((uint32_t)sl.ev[i] & 0xFFu) << 8);
                            en.aff_off = sl.aff[i];
                            en.push_off = NSK_NO_STREAM;
                            en.wide_off = sl.wide[i] >= 0 ? (uint32_t)sl.wide[i] : NSK_NO_STREAM;     // (the descriptors start at the quad of pos0)
                            if (sl.wide[i] >= 0) {
                                const int stride = NSK_WIDE_STRIDE(sl.nch);
                                for (int64_t P = (pos0 + 255) & ~(int64_t)255; P + 256 <= pos0 + 64 * (int64_t)nt_i; P += 256)
                                    if (g->c.seg_wide[(size_t)sl.wide[i] + (size_t)((P >> 8) - (pos0 >> 8)) * stride] != 0xFFFFFFFFu) nwide++;
                            }
                            vt += (nt_i + lead + 3) & ~3;
                        }
                        wt.ntiles = vt;
                        wp.vt = vt;
                        const char *min_env = nsk::diag_env("NSK_WIDE_LEARN_MIN");                 // (diagnostic; the small-grid tests use 0)
                        const int min_quads = min_env ? atoi(min_env) : NSK_WIDE_LEARN_MIN_QUADS;
                        wp.wide = 8 * nwide >= vt && vt > 0 && vt / 4 >= min_quads && !nsk::diag_env("NSK_NO_WIDE_LEARN");
                        if (wp.wide) nsk_tabw_rest_list(g->c, wt, sl.nch, wp.nrest, wp.rest);
                    }
                    if (wp.wide) {
                        const DevGraph<signed char> &dw = d;
                        const bool fuse = SMALLW && pend.valid && pend.tabs_here;
                        const ApplyArgs &prev = fuse ? pend.aa : no_update;
                        TabwRest rest;
                        rest.n = wp.nrest;
                        memcpy(rest.q, wp.rest, sizeof(rest.q));
                        const int grid = nsk_learn_tabw_grid(wp.vt) + (SMALLW ? NSK_SERVICE_BLOCKS : 0) + nsk_tabw_front_blocks(rest.n);
                        if (sl.nch == 1) k_learn_seg_tabw<SMALLW, 1><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(dw, wp.tab, lp, prev, rest);
                        else k_learn_seg_tabw<SMALLW, 2><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(dw, wp.tab, lp, prev, rest);
                        if (fuse) pend.valid = false;
                        g->launches++;
                        continue;
                    }
                }
                // table launches number their tiles virtually: every segment is padded to whole trips
                int vt = 0;
                for (int i = 0; i < NSK_SEG_MAX; i++) {
                    SegEntry &en = tab.e[i];
                    const int nt_i = i < sl.n ? sl.tile_start[i + 1] - sl.tile_start[i] : 0;
                    en.tile_start = use_tab ? vt : (i < sl.n ? sl.tile_start[i] : sl.tile_start[sl.n]);
                    vt += (nt_i + NSK_LEARN_TPW - 1) / NSK_LEARN_TPW * NSK_LEARN_TPW;
                    en.ntiles_lead = (uint32_t)nt_i;
                    en.pos0 = sl.pos0[i]; en.adj_off = sl.adj_off[i]; en.prog = sl.prog[i];
                    en.zoff = sl.zoff[i];
                    en.zmask_ev = (sl.zmask[i] & 0xFFu) | (((uint32_t)sl.ev[i] & 0xFFu) << 8);
                    en.aff_off = use_tab ? sl.aff[i] : NSK_NO_STREAM;         // implicit adjacency (table kernel)
                }
                tab.ntiles = use_tab ? vt : sl.tile_start[sl.n];
                const int grid = nsk_learn_seg_grid(sl, nw, SMALLW, g->values_regular);
#define NSK_LSEG(KIND, NCH) k_learn_seg<VT, SMALLW, KIND, NCH><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(d, tab, lp)
                if (use_tab) {
                    // the previous class's update rides in this launch (block 0 of the service blocks in front)
                    const bool fuse = SMALLW && pend.valid && pend.tabs_here;
                    const ApplyArgs &prev = fuse ? pend.aa : no_update;
                    const int gridx = grid + (SMALLW ? NSK_SERVICE_BLOCKS : 0);
#define NSK_LTAB(NCH) k_learn_seg_tab<VT, SMALLW, NCH, NSK_LEARN_TPW><<<dim3(gridx), dim3(NSK_BLOCK), shmem, g->stream>>>(d, tab, lp, prev)
                    if (sl.nch == 1) NSK_LTAB(1); else NSK_LTAB(2);
#undef NSK_LTAB
                    if (fuse) pend.valid = false;
                }
                else if (sl.tab) { if (sl.nch == 1) NSK_LSEG(0, 1); else NSK_LSEG(0, 2); }   // tables unusable: the
                                                                       // generic slot algebra serves every function
                else if (sl.kind == 4) { if (sl.nch == 1) NSK_LSEG(4, 1); else NSK_LSEG(4, 2); }
                else if (sl.kind == 2) { if (sl.nch == 1) NSK_LSEG(2, 1); else NSK_LSEG(2, 2); }
                else if (sl.kind == 0) { if (sl.nch == 1) NSK_LSEG(0, 1); else NSK_LSEG(0, 2); }
                else { if (sl.nch == 1) NSK_LSEG(3, 1); else NSK_LSEG(3, 2); }
#undef NSK_LSEG
                g->launches++;
            }
            if (nlrest > 0 && !rest_in_general) {   // the other uniform and shape tiles: descriptor-driven kernel
                const int grid = std::min(NSK_LEARN_FAST_BLOCKS, (nlrest + 3) / 4);
                k_learn_fast<VT, SMALLW><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(
                    d, fb, fe, (int)g->c.phase_wb_base[ph], g->learn_rest_tiles + g->c.phase_learn_rest_base[ph],
                    nlrest, lp);
                g->launches++;
            }
            if (ndyn > 0) {             // tiles with per-lane headers: generic kernel, list mode
                const int grid = std::min(NSK_LEARN_LIST_BLOCKS, (ndyn + 3) / 4);
                k_learn_phase<VT, SMALLW, false><<<dim3(grid), dim3(NSK_BLOCK), shmem, g->stream>>>(
                    d, fb, fe, g->dyn_tiles + g->c.phase_dyn_base[ph], ndyn, lp);
                g->launches++;
            }
            cs.join();
            if (nw > 0) {
                if (pend.valid) { launch_update(pend); pend.valid = false; }     // the previous class's, not fused above
                Pending u;
                u.valid = true; u.set = set; u.tabs_here = tabs_here; u.kstat = kstat_here; u.ph = ph; u.step = step; u.aa = aa;
                if (lag) pend = u;              // behind the next class
                else launch_update(u);
            }
            cls++;
        }
        g->sweep++;
        step *= decay;                                   // factorgraph.py:206
    }
    if (pend.valid) { launch_update(pend); pend.valid = false; }
    if (lag && cls > 0 && ((cls - 1) & 1))          // the call leaves the weights, every update applied, in set 0
        HIPCHECK(hipMemcpyAsync(g->w, g->w1, (size_t)nw * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
    g->adj_wt_skip = false;
    g->weights_dirty = true;        // the next inference call rebuilds prog_w and the weight rows
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

#if NSK_LEARN_PART != -1
#if NSK_LEARN_HAS(0)
template int nsk_learn_chromatic<int8_t, false>(NSK_LEARN_SIG);
#endif
#if NSK_LEARN_HAS(1)
template int nsk_learn_chromatic<int8_t, true>(NSK_LEARN_SIG);
#endif
#if NSK_LEARN_HAS(2)
template int nsk_learn_chromatic<int32_t, false>(NSK_LEARN_SIG);
#endif
#if NSK_LEARN_HAS(3)
template int nsk_learn_chromatic<int32_t, true>(NSK_LEARN_SIG);
#endif
#endif

#if NSK_LEARN_PART <= 0         // the sequential validation scan and the entry point
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
        int rc = g->smallw ? nsk_learn_chromatic<VT, true>(g, nsweeps, step, decay, regularization, reg_param,
                                                           truncation, learn_non_evidence)
                           : nsk_learn_chromatic<VT, false>(g, nsweeps, step, decay, regularization, reg_param,
                                                            truncation, learn_non_evidence);
        if (rc) return rc;
    }
    g->sweeps_done += nsweeps;
    return NSK_OK;
}

extern "C" int nsk_learn_sweeps(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                     double reg_param, int64_t truncation, int learn_non_evidence) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    if (regularization == 1 && truncation == 0) return fail(NSK_E_INVALID, "truncation must be non-zero (ZeroDivisionError in the reference)");
    if (nsweeps == 0) return NSK_OK;
    // chromatic learning sums a class's gradients as fixed point (order-free, deterministic): Q31.32, or
    // Q(31+s).(32-s) when the bound on one weight's sum reaches 2^30 (nsk_compile.cpp grad_shift)
    if (g->scan == NSK_SCAN_CHROMATIC && g->c.grad_bound >= 1073741824.0 * 4294967296.0)
        return fail(NSK_E_RANGE, "the gradient sum of one weight in one colour class can exceed a 63-bit integer "
                                 "accumulator (|featureValue| x visits >= 2^62); rescale featureValue or "
                                 "use the sequential scan");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    return g->c.vbytes == 1
               ? learn_impl<int8_t>(g, nsweeps, step, decay, regularization, reg_param, truncation, learn_non_evidence)
               : learn_impl<int32_t>(g, nsweeps, step, decay, regularization, reg_param, truncation, learn_non_evidence);
}
#endif
