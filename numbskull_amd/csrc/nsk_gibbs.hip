// nsk_gibbs.hip -- inference sweep driver: replaces run_pool(gibbsthread) at
// numbskull/factorgraph.py:141 (burn-in) and :163 (inference); kernels in nsk_kernels_gibbs.h.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nsk_internal.h"
#include "nsk_kernels_misc.h"

using namespace nsk;

// hot arguments of k_gibbs_seg_tabw (nsk_kernels_gibbs.h NSK_TABW_HOT): the other arrays as 256-byte units from `val`
static inline int tabw_delta(const void *p, const void *base) {
    const long long d = (const char *)p - (const char *)base;
    return (int)(d / 256);            // (device allocations are 256-byte aligned: checked once in nsk_ensure_seg_plans)
}
#define NSK_TABW_HOT_ARGS(G, T, NB, NF, SB) (signed char *)(G)->val, tabw_delta((G)->cnt_pos, (G)->val), tabw_delta((G)->seg_wide, (G)->val),            \
        tabw_delta((G)->ztab, (G)->val), (T).e[1].tile_start, (T).e[0].ntiles_lead, (T).e[0].pos0, (T).e[0].wide_off, (T).e[0].zoff,             \
        (uint32_t)(T).ntiles | ((uint32_t)((T).n - 1) << 28),                                                                                    \
        ((T).e[0].zmask_ev & 0xFFu) | ((uint32_t)(((NB) >> 3) * (NSK_BLOCK / 64)) << 8) | ((uint32_t)(NF) << 24),                                \
        (const unsigned long long *)(SB)

// The segment launches of every colour, prepared once and kept in the handle (a small graph's sweep is two
// 4 us kernels: rebuilding the tables per sweep made the host the bottleneck): segments batched by (kind,
// chunks) into tables of <= NSK_SEG_MAX; kind 8 = segments with draw tables (any function: the table encodes
// it).  A handle that exchanges its boundary inside the table launches (p2p_fused) gets its segments split
// into runs of border tiles (SegEntry.push_off = their first row in the push map) and runs of interior tiles.
void nsk_ensure_seg_plans(nsk_graph *g, int sample_evidence) {
    typedef NskSegPlan SegPlan;
    const size_t nphase = g->c.phase_start.size() - 1;
    std::vector<std::vector<SegPlan>> &seg_plans = g->seg_plans;
    const int plans_key = (sample_evidence ? 1 : 0) | (g->values_regular ? 2 : 0) | (g->p2p_fused ? 4 : 0);
    if (g->seg_plans_key == plans_key) return;
    g->seg_plans_key = plans_key;
    seg_plans.assign(nphase, std::vector<SegPlan>());
    const bool use_tab = g->values_regular;
    struct Run { const Compiled::Segment *sg; int t0, nt; uint32_t push_off; };
    const std::vector<int32_t> &bt = g->p2p_border_tiles;
    uint32_t border_total = 0;
    bool border_all = true;
    if (g->p2p_fused)          // border tiles of segments this call does not sample: the fused exchange cannot run
        for (const Compiled::Segment &sg : g->c.segments)
            if (!(sg.ev == 0 || sample_evidence)) {
                const int32_t f = (int32_t)(sg.pos0 / 64);
                const auto it = std::lower_bound(bt.begin(), bt.end(), f);
                if (it != bt.end() && *it < f + sg.ntiles) border_all = false;
            }
    int first_phase = -1, last_phase = -1;            // classes with sampled segments
    for (const Compiled::Segment &sg : g->c.segments)
        if (sg.ev == 0 || sample_evidence) {
            if (first_phase < 0 || sg.phase < first_phase) first_phase = sg.phase;
            if (sg.phase > last_phase) last_phase = sg.phase;
        }
    g->p2p_first_phase = first_phase;
    for (size_t ph = 0; ph < nphase; ph++)
        for (int kind = 0; kind <= 8; kind++) {
            if (kind == 1 || (kind > 4 && kind < 8)) continue;   // IMPLY_NATURAL shares the AND step (3)
            for (int nch = 1; nch <= 2; nch++) {
                SegTable tab;
                memset(&tab, 0, sizeof(tab));
                auto flush = [&]() {
                    if (tab.n == 0) return;
                    for (int i = tab.n; i < NSK_SEG_MAX; i++) tab.e[i].tile_start = tab.ntiles;
                    // the wide-quad kernel takes the launch when at least half of its quads are wide ones (it samples
                    // the others one tile at a time)
                    const auto fits = [&](const void *p) {         // (the kernel's hot arguments: 256-byte units from val in 32 bits)
                        const long long d = (const char *)p - (const char *)g->val;
                        return d % 256 == 0 && d / 256 > -(1ll << 31) && d / 256 < (1ll << 31);
                    };
                    tab.wide = (kind >= 8 && 8 * tab.wide >= tab.ntiles && tab.ntiles < (1 << 28) && fits(g->cnt_pos) && fits(g->seg_wide) &&
                                fits(g->ztab) && !nsk::diag_env("NSK_NO_WIDE_KERNEL")) ? 1 : 0;
                    SegPlan pl{kind, nch, tab};
                    if (tab.wide) nsk_tabw_rest_list(g->c, pl.tab, pl.nch, pl.nrest, pl.rest);
                    seg_plans[ph].push_back(pl);
                    memset(&tab, 0, sizeof(tab));
                };
                // largest segments first: the kernels find a tile's segment with a scan
                // whose first probe is the table's first entry
                std::vector<Run> mine;
                for (const Compiled::Segment &sg : g->c.segments) {
                    if (sg.phase != (int)ph) continue;
                    const int k3 = (use_tab && sg.ztab >= 0) ? 8 : sg.kind == 1 ? 3 : (int)sg.kind;
                    if (k3 != kind || (sg.nslots > 4 ? 2 : 1) != nch) continue;
                    if (!(sg.ev == 0 || sample_evidence)) continue;      // inference.py:24
                    if (!g->p2p_fused || kind < 8) { mine.push_back(Run{&sg, 0, sg.ntiles, NSK_NO_STREAM}); continue; }
                    // runs of border / interior tiles (a tile's rank among the border tiles = its push-map row)
                    const int32_t f = (int32_t)(sg.pos0 / 64);
                    size_t bi = (size_t)(std::lower_bound(bt.begin(), bt.end(), f) - bt.begin());
                    for (int t = 0; t < sg.ntiles;) {
                        const bool isb = bi < bt.size() && bt[bi] == f + t;
                        int e = t + 1;
                        if (isb) { while (e < sg.ntiles && bi + (size_t)(e - t) < bt.size() && bt[bi + (size_t)(e - t)] == f + e) e++; }
                        else e = (bi < bt.size() && bt[bi] < f + sg.ntiles) ? bt[bi] - f : sg.ntiles;
                        mine.push_back(Run{&sg, t, e - t, isb ? (uint32_t)bi : NSK_NO_STREAM});
                        if (isb) { bi += (size_t)(e - t); border_total += (uint32_t)(e - t); }
                        t = e;
                    }
                }
                // largest first -- but, for a shard that exchanges inside its launches, the border runs in front: their
                // chain (flag, system-coherent ghost loads, pushes, acknowledgement, counter) is twice a trip long and
                // overlaps the interior trips when it starts first; behind them it was a tail of every launch (a
                // border wave that waits for a peer holds one of 6144 wave slots, nothing else)
                (void)last_phase;
                std::stable_sort(mine.begin(), mine.end(), [&](const Run &a, const Run &b) {
                    const bool ba = a.push_off != NSK_NO_STREAM, bb = b.push_off != NSK_NO_STREAM;
                    if (ba != bb) return ba;
                    return a.nt > b.nt; });
                for (const Run &rn : mine) {
                    const Compiled::Segment &sg = *rn.sg;
                    SegEntry &en = tab.e[tab.n];
                    const int64_t pos0 = sg.pos0 + 64 * (int64_t)rn.t0;
                    // table launches number their tiles virtually: a segment starts on a quad
                    // boundary (positions 256 m), with up to three dead tiles in front
                    const int lead = kind >= 8 ? (int)((pos0 / 64) & 3) : 0;
                    const int vtiles = kind >= 8 ? ((rn.nt + lead + 3) & ~3) : rn.nt;
                    en.ntiles_lead = (uint32_t)rn.nt | ((uint32_t)lead << 30);
                    en.tile_start = tab.ntiles;
                    en.pos0 = (int)pos0;
                    en.adj_off = sg.adj_off + (uint32_t)rn.t0 * 64u * (uint32_t)nch;
                    en.prog = sg.prog;
                    en.zoff = sg.ztab >= 0 ? (uint32_t)sg.ztab : 0u;
                    // (bit 16: the positions of a segment with a draw table draw from the quad
                    // scheme whichever kernel samples them, nsk_device.h quad_block)
                    en.zmask_ev = ((1u << sg.nslots) - 1u) | (((uint32_t)sg.ev & 0xFFu) << 8) | (sg.ztab >= 0 ? 1u << 16 : 0u);
                    en.aff_off = (kind >= 8 && sg.aff >= 0) ? (uint32_t)sg.aff + (uint32_t)rn.t0 * (uint32_t)nch : NSK_NO_STREAM;
                    en.push_off = rn.push_off;
                    // quad descriptors (nsk_compile.h seg_wide) from the quad of the run's first position on; NCH of the
                    // descriptors = the segment's own chunk count = this launch's
                    en.wide_off = (sg.ztab >= 0 && sg.wide >= 0)
                        ? (uint32_t)(sg.wide + ((pos0 >> 8) - (sg.pos0 >> 8)) * NSK_WIDE_STRIDE(nch)) : NSK_NO_STREAM;
                    if (en.wide_off != NSK_NO_STREAM) {      // the wide quads the run touches; those it holds whole (SegTable.wide, flush)
                        const int stride = NSK_WIDE_STRIDE(nch);
                        int touched = 0;
                        for (int64_t P = pos0 & ~(int64_t)255; P < pos0 + 64 * (int64_t)rn.nt; P += 256)
                            if (g->c.seg_wide[(size_t)sg.wide + (size_t)((P >> 8) - (sg.pos0 >> 8)) * stride] != 0xFFFFFFFFu) {
                                touched++;
                                if (kind >= 8 && P >= pos0 && P + 256 <= pos0 + 64 * (int64_t)rn.nt) tab.wide++;
                            }
                        if (!touched) en.wide_off = NSK_NO_STREAM;       // (none: the kernels need not look)
                    }
                    tab.ntiles += vtiles;
                    if (++tab.n == NSK_SEG_MAX) flush();
                }
                flush();
            }
        }
    // The border waves of the sweep's first class THAT HAS BORDER TILES wait for the peers' flags of the previous
    // exchange (later classes find them raised: the stream is in order).  Not simply the first class with sampled
    // segments: a cut whose boundary variables all have the other colour (a chain, a bipartite cut) has no border tile
    // there, and then nothing in the sweep would wait -- ghosts read before they arrived, pushes over a block a peer
    // still reads.
    if (g->p2p_fused) {
        int first_border = -1;
        for (size_t ph = 0; ph < nphase && first_border < 0; ph++)
            for (const SegPlan &pl : seg_plans[ph])
                for (int i = 0; i < pl.tab.n && first_border < 0; i++)
                    if (pl.tab.e[i].push_off != NSK_NO_STREAM) first_border = (int)ph;
        if (first_border >= 0) g->p2p_first_phase = first_border;
    }
    g->p2p_border_total = border_total;
    g->p2p_border_all = border_all && border_total == (uint32_t)bt.size();
}

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
        nsk_refresh_prog_weights(g);
        nsk_ensure_seg_plans(g, sample_evidence);
        std::vector<std::vector<NskSegPlan>> &seg_plans = g->seg_plans;
        typedef NskSegPlan SegPlan;
        for (int64_t s = 0; s < nsweeps; s++) {
            if (g->pack_now && !burnin && g->packed_sweeps >= 127) (void)nsk_unpack_tally(g);       // 7 tally bits per value byte
            if (g->p2p_fused_now) ++g->p2p_tag;            // the exchange rides in this sweep's table launches
            for (size_t ph = 0; ph < nphase; ph++) {
                const int fb = (int)g->c.phase_start[ph], fe = (int)g->c.phase_fast_end[ph];
                const int e = (int)g->c.phase_end[ph];
                const int he = (int)g->c.phase_heavy_end[ph];
                ColourStreams cs(g, !g->no_overlap);
                bool rest_in_general = false;       // the colour's rest tiles were given to a general launch
                const bool ep = fe > fb && g->c.phase_ep[ph];
                if (ep) {   // entry-parallel groups: hubs, the colour's general tiles and its rest tiles in one launch
                    const int gt0 = (int)g->c.phase_gen_tile[ph];
                    const int ngt = (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]) - gt0;
                    const int ngroups = (int)(g->c.phase_ep_base[ph + 1] - g->c.phase_ep_base[ph]);
                    const int nbh = (int)(g->c.phase_bighub_base[ph + 1] - g->c.phase_bighub_base[ph]);   // a block per long-list hub
                    const int hblocks = nbh + (he - fe + 3) / 4;
                    const int nrest_all = (int)(g->c.phase_rest_base[ph + 1] - g->c.phase_rest_base[ph]);
                    rest_in_general = nrest_all > 0;
                    const int rblocks = (nrest_all + 3) / 4;
                    // grid: 7 workgroups per CU, whole rounds of XCDs.  Five of them are resident (90 / 86 vector
                    // registers since the wave index is a scalar; the categorical kernel had 101 and four until round 5); the
                    // others start as those end, which deals the
                    // groups' uneven costs out dynamically -- per class on the 5M LR graph (NSK_EP_PER_CU):
                    // 3 workgroups per CU 68.4 us, 4 58.6, 5 65.6, 6 66.8, 7 58.3
                    const bool cat8 = g->c.phase_gen_bin_tile[ph] > gt0;
                    const char *pcu_env = nsk::diag_env("NSK_EP_PER_CU");        // (diagnostic: workgroups per CU)
                    const int pcu = pcu_env ? std::max(1, std::min(32, atoi(pcu_env))) : 7;
                    const int gblocks = 8 * ((std::min(ngroups, 256 * pcu) + 7) / 8);
                    const dim3 grid(hblocks + gblocks + rblocks);
                    const size_t smem = 0;
#define NSK_EP_ARGS d, fb, fe, (int)g->c.phase_wb_base[ph], gt0, ngt, ngroups, (int)g->c.phase_ep_base[ph], gblocks, fe, he, hblocks, \
                    (int)g->c.phase_hub_base[ph], nbh, (int)g->c.phase_bighub_base[ph], g->rest_tiles + g->c.phase_rest_base[ph], nrest_all, sample_evidence, burnin, \
                    (uint32_t)g->seed, (uint32_t)(g->seed >> 32), (uint32_t)g->sweep, nsk_sweep_hi(g)
                    // (value array within the L2s: the register-capped twin with a fifth wave per SIMD)
                    const bool in_l2 = (size_t)g->c.nid * (size_t)g->c.vbytes <= ((size_t)24 << 20);
                    if (cat8 && in_l2) k_gibbs_ep_w5<VT, 8><<<grid, dim3(NSK_BLOCK), smem, g->stream>>>(NSK_EP_ARGS);
                    else if (cat8) k_gibbs_ep<VT, 8><<<grid, dim3(NSK_BLOCK), smem, g->stream>>>(NSK_EP_ARGS);
                    else k_gibbs_ep<VT, 2><<<grid, dim3(NSK_BLOCK), smem, g->stream>>>(NSK_EP_ARGS);
#undef NSK_EP_ARGS
                    g->launches++;
                }
                if (!ep) {   // hubs (one wave per variable) + general tiles with categorical lanes: one launch
                    const int gt0 = fe > fb ? (int)g->c.phase_gen_tile[ph] : 0;
                    int gtb = fe > fb ? (int)g->c.phase_gen_bin_tile[ph] : 0;
                    // a class with categorical tiles walks ALL its general tiles in this launch, on
                    // the main stream: the fork / join events of a side stream cost more (~20 us per
                    // class) than the binary tiles lose by running the 8-candidate code
                    const bool one_general = gtb > gt0;
                    if (one_general) gtb = (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]);
                    // without categorical tiles the hubs ride in the binary launch on the main stream
                    // (no side stream, no fork / join events for this class)
                    const bool hubs_with_binary = gtb == gt0 && fe > fb &&
                        (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]) > gtb;
                    const int nblocks = (gtb - gt0 + 3) / 4, hblocks = hubs_with_binary ? 0 : (he - fe + 3) / 4;
                    // the colour's other tiles outside segments ride in the general launch too
                    const int nrest_all = fe > fb ? (int)(g->c.phase_rest_base[ph + 1] - g->c.phase_rest_base[ph]) : 0;
                    const bool rest_here = one_general && nrest_all > 0;
                    rest_in_general = rest_here;
                    const int rblocks = rest_here ? (nrest_all + 3) / 4 : 0;
                    if (nblocks + hblocks > 0) {
                        k_gibbs_general<VT, 8><<<dim3(hblocks + 8 * ((nblocks + 7) / 8) + rblocks), dim3(NSK_BLOCK), 0, one_general ? g->stream : cs.side(0)>>>(
                            d, fb, fe, (int)g->c.phase_wb_base[ph], gt0, gtb - gt0, nblocks, fe, he, hblocks, (int)g->c.phase_hub_base[ph],
                            g->rest_tiles + g->c.phase_rest_base[ph], rest_here ? nrest_all : 0,
                            sample_evidence, burnin, (uint32_t)g->seed, (uint32_t)(g->seed >> 32),
                            (uint32_t)g->sweep, nsk_sweep_hi(g));
                        g->launches++;
                    }
                }
                if (e > he) {       // generic CSR kernel, one lane per variable
                    k_gibbs_phase<VT><<<dim3((e - he + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, cs.side(1)>>>(
                        d, he, e, sample_evidence, burnin, (uint32_t)g->seed, (uint32_t)(g->seed >> 32),
                        (uint32_t)g->sweep, nsk_sweep_hi(g));
                    g->launches++;
                }
                if (fe > fb) {      // inlined-adjacency kernels
                    const uint32_t K0 = (uint32_t)g->seed, K1 = (uint32_t)(g->seed >> 32);
                    const uint32_t S0 = (uint32_t)g->sweep, S1 = nsk_sweep_hi(g);
                    const int gt0 = (int)g->c.phase_gen_tile[ph];
                    const int ngt = (int)(g->c.phase_wb_base[ph + 1] - g->c.phase_wb_base[ph]) - gt0;
                    int gtb = (int)g->c.phase_gen_bin_tile[ph];
                    if (gtb > gt0) gtb = gt0 + ngt;     // walked by the launch above
                    if (gt0 + ngt > gtb && !ep) {   // all-binary general tiles (IMPLY_MLN, mixed tails)
                        const int nblocks = (gt0 + ngt - gtb + 3) / 4;
                        const int hbl = gtb == gt0 ? (he - fe + 3) / 4 : 0;     // see above
                        const int nrest_all = (int)(g->c.phase_rest_base[ph + 1] - g->c.phase_rest_base[ph]);
                        rest_in_general = gtb == gt0 && nrest_all > 0;
                        const int rblocks = rest_in_general ? (nrest_all + 3) / 4 : 0;
                        k_gibbs_general<VT, 2><<<dim3(hbl + 8 * ((nblocks + 7) / 8) + rblocks), dim3(NSK_BLOCK), 0, g->stream>>>(
                            d, fb, fe, (int)g->c.phase_wb_base[ph], gtb, gt0 + ngt - gtb, nblocks, fe, he, hbl, (int)g->c.phase_hub_base[ph],
                            g->rest_tiles + g->c.phase_rest_base[ph], rest_in_general ? nrest_all : 0,
                            sample_evidence, burnin, K0, K1, S0, S1);
                        g->launches++;
                    }
                    // segments of this colour: the launches prepared before the sweep loop
                    for (const SegPlan &pl : seg_plans[ph]) {
                        const SegTable &tab = pl.tab;
                        const int kind = pl.kind, nch = pl.nch;
                        const int nb = (tab.ntiles + 3) / 4;
                        const dim3 grid(8 * ((nb + 7) / 8)), block(NSK_BLOCK);
#define NSK_SEG(KIND, NCH) k_gibbs_seg<VT, KIND, NCH><<<grid, block, 0, g->stream>>>(d, tab, nb, burnin, K0, K1, S0, S1)
                        if (kind >= 8) {
                            // a wave per tile pair while that fits the resident grid, else its waves loop over quads
                            const int nbp = nsk_tab_grid(tab.ntiles);
                            if (g->p2p_fused_now) {
                                TabP2P px;
                                nsk_p2p_fill(g, px, nullptr, g->p2p_tag, (int)ph == g->p2p_first_phase);
                                if (nch == 1) k_gibbs_seg_tab_p2p<VT, 1><<<dim3(nbp), block, 0, g->stream>>>(d, tab, burnin, K0, K1, S0, S1, nullptr, 0u, px);
                                else k_gibbs_seg_tab_p2p<VT, 2><<<dim3(nbp), block, 0, g->stream>>>(d, tab, burnin, K0, K1, S0, S1, nullptr, 0u, px);
                            }
                            else if (sizeof(VT) == 1 && tab.wide) {          // (mostly) wide quads: four positions to a lane
                                const DevGraph<signed char> dw = view<signed char>(g);
                                const int nbw = nsk_tabw_grid(tab.ntiles);
                                const int mode = burnin ? 1 : (g->pack_now ? 2 : 0);        // 2: the tally inside the value bytes
                                TabwCold cold{K0, K1, S0, S1, 0u, 0u, dw, tab, pl.nrest, {}};
                                memcpy(cold.rest, pl.rest, sizeof(cold.rest));
                                const int nf = nsk_tabw_front_blocks(pl.nrest);
#define NSK_TABW(NCH, MODE) k_gibbs_seg_tabw<NCH, MODE><<<dim3(nf + nbw), block, 0, g->stream>>>(NSK_TABW_HOT_ARGS(g, tab, nbw, nf, nullptr), cold)
                                if (nch == 1) { if (mode == 0) NSK_TABW(1, 0); else if (mode == 1) NSK_TABW(1, 1); else NSK_TABW(1, 2); }
                                else { if (mode == 0) NSK_TABW(2, 0); else if (mode == 1) NSK_TABW(2, 1); else NSK_TABW(2, 2); }
#undef NSK_TABW
                            }
                            else if (nch == 1) k_gibbs_seg_tab<VT, 1><<<dim3(nbp), block, 0, g->stream>>>(d, tab, burnin, K0, K1, S0, S1, nullptr, 0u);
                            else k_gibbs_seg_tab<VT, 2><<<dim3(nbp), block, 0, g->stream>>>(d, tab, burnin, K0, K1, S0, S1, nullptr, 0u);
                        }
                        else if (kind == 4) { if (nch == 1) NSK_SEG(4, 1); else NSK_SEG(4, 2); }
                        else if (kind == 2) { if (nch == 1) NSK_SEG(2, 1); else NSK_SEG(2, 2); }
                        else if (kind == 0) { if (nch == 1) NSK_SEG(0, 1); else NSK_SEG(0, 2); }
                        else { if (nch == 1) NSK_SEG(3, 1); else NSK_SEG(3, 2); }
#undef NSK_SEG
                        g->launches++;
                    }
                    const int nrest = (int)(g->c.phase_rest_base[ph + 1] - g->c.phase_rest_base[ph]);
                    if (nrest > 0 && !rest_in_general) {
                        const int nblocks = (nrest + 3) / 4;
                        k_gibbs_fast<VT><<<dim3(8 * ((nblocks + 7) / 8)), dim3(NSK_BLOCK), 0, g->stream>>>(
                            d, fb, fe, (int)g->c.phase_wb_base[ph], nblocks,
                            g->rest_tiles + g->c.phase_rest_base[ph], nrest, sample_evidence, burnin,
                            K0, K1, S0, S1);
                        g->launches++;
                    }
                }
                cs.join();
            }
            g->sweep++;
            if (!burnin && g->pack_now) g->packed_sweeps++;
            if (!burnin && ++g->pos_tally_sweeps == 255) nsk_fold_position_tally(g);   // uint8 tally is full
        }
        HIPCHECK(hipGetLastError());
    }
    if (!burnin) g->cnt_dirty = true;
    g->sweeps_done += nsweeps;
    return NSK_OK;
}


#ifdef NSK_ABL_TIMING       // (instrumented build: the per-wave time stamps of the last wide-quad launch, tools/timing_tabw.py)
extern "C" int nsk_debug_dump(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk::nsk_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif

static int gibbs_eager(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    return g->c.vbytes == 1 ? gibbs_impl<int8_t>(g, nsweeps, sample_evidence, burnin)
                            : gibbs_impl<int32_t>(g, nsweeps, sample_evidence, burnin);
}

// ---- captured sweep sequences (hipGraph) -------------------------------------------------------------
// A handle whose inference sweep consists of table launches only (grids and their shards: two ~4 us
// kernels per sweep, plus two exchange kernels in an N-rank run) is bound by launches, not by kernels.
// NSK_GRAPH_SWEEPS sweeps -- class launches, peer-to-peer push and wait/unpack -- are captured once into
// a hipGraph whose kernels read the sweep index (and the exchange tag) from device memory + a per-node
// offset, so the same executable graph serves every replay; a one-thread kernel at its end advances the
// counters.  The uint8 position tally is folded between replays when it would overflow.
static bool graph_eligible(const nsk_graph *g, bool p2p) {
    if (g->scan != NSK_SCAN_CHROMATIC || !g->values_regular || nsk::diag_env("NSK_NO_GRAPH")) return false;
    const nsk::Compiled &c = g->c;
    // launches set the pace only while the class kernels are short: beyond a few million variables per
    // handle (10M grid: 12 us per class) a replay saves nothing and its launch latency shows in short runs.
    // A handle that exchanges peer to peer adds two tiny kernels per sweep: its sequences are captured up
    // to twice the size (the two shards of the 10M grid)
    if (c.nsampled > (p2p ? 6000000 : 3000000)) return false;
    return nsk_tables_only(g);
}

template <typename VT>
static int graph_build(nsk_graph *g, int sample_evidence, int burnin, bool p2p, int key, bool big) {
    hipGraphExec_t &exec = big ? g->sweep_graph_big : g->sweep_graph;
    int &exec_key = big ? g->sweep_graph_big_key : g->sweep_graph_key;
    int &exec_launches = big ? g->sweep_graph_big_launches : g->sweep_graph_launches;
    const int nsw = big ? NSK_GRAPH_SWEEPS_BIG : NSK_GRAPH_SWEEPS;
    if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
    exec_key = -1;
    // the segment plans (kept in the handle) are built by an eager sweep-free call path: make sure they exist
    DevGraph<VT> d = view<VT>(g);
    hipGraph_t graph = nullptr;
    // (the legacy default stream cannot be captured: a caller that pointed the library at it -- torch's
    // current stream in a process without its own streams -- keeps the eager loop)
    if (g->stream == nullptr) return nsk::fail(NSK_E_DEVICE, "the default stream cannot be captured");
    HIPCHECK(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
    int launches = 0;
    for (int i = 0; i < nsw; i++) {
        for (size_t ph = 0; ph < g->seg_plans.size(); ph++)
            for (const NskSegPlan &pl : g->seg_plans[ph]) {
                const int nbp = nsk_tab_grid(pl.tab.ntiles);
                if (g->p2p_fused_now) {          // the exchange inside the launch: tag = counter + i + 1
                    TabP2P px;
                    nsk_p2p_fill(g, px, g->d_counters, (unsigned int)(i + 1), (int)ph == g->p2p_first_phase);
                    if (pl.nch == 1)
                        k_gibbs_seg_tab_p2p<VT, 1><<<dim3(nbp), dim3(NSK_BLOCK), 0, g->stream>>>(d, pl.tab, burnin, 0u, 0u, 0u, 0u,
                                                                                                 g->d_counters, (uint32_t)i, px);
                    else
                        k_gibbs_seg_tab_p2p<VT, 2><<<dim3(nbp), dim3(NSK_BLOCK), 0, g->stream>>>(d, pl.tab, burnin, 0u, 0u, 0u, 0u,
                                                                                                 g->d_counters, (uint32_t)i, px);
                }
                else if (sizeof(VT) == 1 && pl.tab.wide) {
                    const DevGraph<signed char> dw = view<signed char>(g);
                    const int nbw = nsk_tabw_grid(pl.tab.ntiles);
                    const int mode = burnin ? 1 : (g->pack_now ? 2 : 0);
                    TabwCold cold{0u, 0u, 0u, 0u, (uint32_t)i, 0u, dw, pl.tab, pl.nrest, {}};
                    memcpy(cold.rest, pl.rest, sizeof(cold.rest));
                    const int nf = nsk_tabw_front_blocks(pl.nrest);
#define NSK_TABW(NCH, MODE) k_gibbs_seg_tabw<NCH, MODE><<<dim3(nf + nbw), dim3(NSK_BLOCK), 0, g->stream>>>(NSK_TABW_HOT_ARGS(g, pl.tab, nbw, nf, g->d_counters), cold)
                    if (pl.nch == 1) { if (mode == 0) NSK_TABW(1, 0); else if (mode == 1) NSK_TABW(1, 1); else NSK_TABW(1, 2); }
                    else { if (mode == 0) NSK_TABW(2, 0); else if (mode == 1) NSK_TABW(2, 1); else NSK_TABW(2, 2); }
#undef NSK_TABW
                }
                else if (pl.nch == 1)
                    k_gibbs_seg_tab<VT, 1><<<dim3(nbp), dim3(NSK_BLOCK), 0, g->stream>>>(d, pl.tab, burnin, 0u, 0u, 0u, 0u,
                                                                                         g->d_counters, (uint32_t)i);
                else
                    k_gibbs_seg_tab<VT, 2><<<dim3(nbp), dim3(NSK_BLOCK), 0, g->stream>>>(d, pl.tab, burnin, 0u, 0u, 0u, 0u,
                                                                                         g->d_counters, (uint32_t)i);
                launches++;
            }
        if (p2p && !g->p2p_fused_now) {
            int rc = nsk_p2p_enqueue(g, g->d_counters, (unsigned int)(i + 1));
            if (rc) { (void)hipStreamEndCapture(g->stream, &graph); if (graph) (void)hipGraphDestroy(graph); return rc; }
        }
    }
    k_graph_counters<<<dim3(1), dim3(1), 0, g->stream>>>(g->d_counters, (unsigned long long)nsw, p2p ? (unsigned long long)nsw : 0ull, 0, 0ull, 0ull);
    hipError_t e = hipStreamEndCapture(g->stream, &graph);
    if (e != hipSuccess || !graph) return nsk::fail(NSK_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { exec = nullptr; return nsk::fail(NSK_E_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    exec_key = key;
    exec_launches = launches;
    (void)sample_evidence;
    return NSK_OK;
}

int nsk_gibbs_run(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin, bool p2p) {
    int64_t left = nsweeps;
    // A shard whose sweep is table launches only exchanges its boundary INSIDE them (nsk_internal.h p2p_fused): the
    // ghosts the first sweep reads are packed into the receive block here, the wait for the peers' last flags and
    // the unpack into the value array are enqueued lazily (nsk_p2p_flush)
    bool fuse = false;
    if (p2p && g->p2p_fused && g->scan == NSK_SCAN_CHROMATIC && g->values_regular && nsweeps > 0) {
        nsk_ensure_seg_plans(g, sample_evidence);
        fuse = g->p2p_border_all;
        for (const auto &v : g->seg_plans) for (const NskSegPlan &pl : v) fuse = fuse && pl.kind >= 8;
        // (a fused call right behind a fused call continues it: the receive block already holds what the first
        // sweep reads -- nothing to unpack into the value array and pack back)
        if (fuse && !g->p2p_close_pending) {
            int rc = nsk_p2p_ghost_pack(g);
            if (rc) return rc;
        }
    }
    if (!fuse) { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    g->p2p_close_pending = false;
    g->p2p_fused_now = fuse;
    // Packed tally (nsk_internal.h): a whole-graph handle whose every launch of this call is the wide-quad kernel's keeps
    // the tally inside the value bytes while the call runs; it is unpacked before the call returns
    g->pack_now = false;
    if (!p2p && !burnin && g->scan == NSK_SCAN_CHROMATIC && g->values_regular && g->c.vbytes == 1 && nsweeps > 0 &&
        nsk_tables_only(g) && !nsk::diag_env("NSK_NO_PACK_TALLY")) {
        nsk_ensure_seg_plans(g, sample_evidence);
        bool all_wide = true;
        for (const auto &v : g->seg_plans) for (const NskSegPlan &pl : v) all_wide = all_wide && pl.kind >= 8 && pl.tab.wide;
        g->pack_now = all_wide;
    }
    struct Done { nsk_graph *g; bool fuse; ~Done() { g->p2p_fused_now = false; if (fuse) g->p2p_close_pending = true;
                                                      (void)nsk_unpack_tally(g); g->pack_now = false; } } done{g, fuse};
    if (left >= NSK_GRAPH_SWEEPS && graph_eligible(g, p2p)) {
        // the plans of this (sample_evidence, tables) combination and the tables of the current weights (what an eager
        // sweep does in front of its launches; an eager sweep in front of the replays cost a 400-sweep call of the 1M grid
        // 16 sweeps outside them -- this one and the 15 that 399 leaves over)
        int rc = NSK_OK;
        nsk_refresh_prog_weights(g);
        nsk_ensure_seg_plans(g, sample_evidence);
        bool all_tab = true;
        for (const auto &v : g->seg_plans) for (const NskSegPlan &pl : v) all_tab = all_tab && pl.kind >= 8;
        const int key = g->seg_plans_key | (burnin ? 8 : 0) | (p2p ? 16 : 0) | (fuse ? 32 : 0) | (g->pack_now ? 64 : 0);
        if (all_tab && left >= NSK_GRAPH_SWEEPS) {
            // two sizes: NSK_GRAPH_SWEEPS_BIG sweeps per replay while the call is long enough, NSK_GRAPH_SWEEPS for what is left
            for (int big = (left >= NSK_GRAPH_SWEEPS_BIG && !nsk::diag_env("NSK_NO_BIG_GRAPH")) ? 1 : 0; big >= 0; big--) {
                const int nsw = big ? NSK_GRAPH_SWEEPS_BIG : NSK_GRAPH_SWEEPS;
                hipGraphExec_t &exec = big ? g->sweep_graph_big : g->sweep_graph;
                int &exec_key = big ? g->sweep_graph_big_key : g->sweep_graph_key;
                if (left < nsw || g->sweep_graph_off) continue;
                if (exec_key != key) {
                    rc = g->c.vbytes == 1 ? graph_build<int8_t>(g, sample_evidence, burnin, p2p, key, big != 0)
                                          : graph_build<int32_t>(g, sample_evidence, burnin, p2p, key, big != 0);
                    if (rc) {               // capture is an optimisation: without it the eager loop runs
                        g->sweep_graph_off = true;
                        (void)hipGetLastError();
                        continue;
                    }
                }
                k_graph_counters<<<dim3(1), dim3(1), 0, g->stream>>>(g->d_counters, g->sweep, g->p2p_tag, 1, g->seed, g->rng_tag);
                while (left >= nsw && exec_key == key) {
                    if (!burnin && g->pos_tally_sweeps + nsw > 255) nsk_fold_position_tally(g);   // uint8 tally
                    if (g->pack_now && g->packed_sweeps + nsw > 127) (void)nsk_unpack_tally(g);   // 7 bits in a value byte
                    HIPCHECK(hipGraphLaunch(exec, g->stream));
                    g->sweep += (uint64_t)nsw;
                    if (p2p) g->p2p_tag += (unsigned int)nsw;
                    if (!burnin) { g->pos_tally_sweeps += nsw; g->cnt_dirty = true; if (g->pack_now) g->packed_sweeps += nsw; }
                    g->sweeps_done += nsw;
                    g->launches += big ? g->sweep_graph_big_launches : g->sweep_graph_launches;
                    left -= nsw;
                }
            }
            HIPCHECK(hipGetLastError());
        }
    }
    for (; left > 0; left--) {              // the rest eagerly (one sweep at a time when exchanging with kernels of their own)
        int rc = gibbs_eager(g, (p2p && !fuse) ? 1 : left, sample_evidence, burnin);
        if (rc) return rc;
        if (!p2p || fuse) break;
        if ((rc = nsk_p2p_enqueue(g, nullptr, 0))) return rc;
    }
    return NSK_OK;
}

extern "C" int nsk_gibbs_sweeps(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    if (nsweeps == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(g->device));
    return nsk_gibbs_run(g, nsweeps, sample_evidence, burnin, false);
}
