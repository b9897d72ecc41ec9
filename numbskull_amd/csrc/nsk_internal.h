// nsk_internal.h -- host-side state of a compiled graph handle, shared by the translation units of
// the library (nsk_api.hip: C-ABI, state sync, exchange; nsk_gibbs.hip / nsk_learn.hip: the sweep
// drivers of numbskull/factorgraph.py:141,163,202).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"
#include "nsk_compile.h"
#include "nsk_device.h"
#include "nsk_kernels_gibbs.h"

namespace nsk {
int fail(int code, const std::string &msg);       // records nsk_last_error, returns code
}

#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return nsk::fail(NSK_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct NskSegPlan { int kind, nch; nsk::SegTable tab;         // one prepared segment launch (nsk_gibbs.hip)
                    uint32_t nrest = 0, rest[NSK_TABW_REST_MAX] = {}; };      // wide launches: the quads that are not wide (TabwCold.rest)

// a learning launch over (mostly) wide quads, prepared once (nsk_learn.hip): its segment table in whole quads, whether the
// wide kernel takes it, and its quads that are not wide ones
struct NskLearnWidePlan { int key = -1; bool wide = false; int vt = 0; nsk::SegTable tab; uint32_t nrest = 0, rest[NSK_TABW_REST_MAX] = {}; };

struct nsk_graph {
    nsk::Compiled c;
    std::vector<NskLearnWidePlan> learn_wide_plans;      // per Compiled::learn_seg entry
    // the inference sweep's segment launches per colour, kept across calls (the N-rank loops sweep one
    // epoch per call); key = sample_evidence | draw tables usable << 1
    std::vector<std::vector<NskSegPlan>> seg_plans;
    int seg_plans_key = -1;
    // captured sweep sequence (hipGraph): NSK_GRAPH_SWEEPS inference sweeps of a handle whose sweep is
    // table launches only (+ the peer-to-peer exchange), replayed with the sweep index in device memory
    hipGraphExec_t sweep_graph = nullptr;
    int sweep_graph_key = -1, sweep_graph_launches = 0;
    // ... and NSK_GRAPH_SWEEPS_BIG of them for long calls (the one-thread kernel that advances the counters and the
    // replay's own latency are ~5 us per replay: 4 % of a 16-sweep replay of the 1M grid)
    hipGraphExec_t sweep_graph_big = nullptr;
    int sweep_graph_big_key = -1, sweep_graph_big_launches = 0;
    bool sweep_graph_off = false;                  // capture failed once (e.g. the legacy default stream): eager from then on
    unsigned long long *d_counters = nullptr;      // [0] sweep index, [1] exchange tag, [2] Philox key, [3] shard tag
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // the kernels of one colour class are independent: hubs and generic-path variables run on side
    // streams next to the tile kernels (fork/join with events around every colour)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool no_overlap = nsk::diag_env("NSK_NO_OVERLAP") != nullptr;     // diagnostic: one stream
    std::vector<void *> allocs;
    std::vector<std::pair<void *, void *>> alloc_alias;   // (aligned pointer handed out, allocation) where they differ (NSK_ALLOC_ALIGN)
    int64_t device_bytes = 0;
    // device arrays
    int32_t *p_vid = nullptr, *p_slot = nullptr, *p_cnt = nullptr, *slot_off = nullptr, *fidx = nullptr;
    uint32_t *p_info = nullptr, *f_rec = nullptr;
    void *p_init = nullptr;
    int32_t *iid_of_vid = nullptr;
    int32_t *m_rec = nullptr, *v_card = nullptr,
            *v_pos = nullptr;
    double *f_feat = nullptr, *w = nullptr, *logtab = nullptr;
    uint8_t *w_fixed = nullptr;
    uint32_t *w_direct = nullptr;      // bit per weight: updated in place by the learning kernels (nsk_compile.h)
    int32_t *multi_wids = nullptr;     // the other weights: what k_apply_weights walks then
    void *val = nullptr, *val_evid = nullptr;
    int32_t *cnt = nullptr;
    uint8_t *cnt_pos = nullptr;
    int pos_tally_sweeps = 0;      // sweeps accumulated in the uint8 position tally
    // Packed tally (k_gibbs_seg_tabw): while a nsk_gibbs_sweeps call runs on a handle whose every launch is the wide-quad
    // kernel's, the tally of its sweeps lives in the value bytes (bit 0 value, bits 1-7 count): one store per trip
    // instead of two.  packed_sweeps = tallied sweeps since the last k_unpack_tally (at most 127; 0 whenever the
    // library returns to the caller: every other reader of values sees plain 0 / 1).
    int packed_sweeps = 0;
    bool pack_now = false;         // the running call sweeps in packed mode
    uint32_t *adj = nullptr, *tiles = nullptr, *tile_hdr = nullptr, *gstream = nullptr, *gs_off = nullptr;
    double *prog_w = nullptr, *adj_wt = nullptr;
    uint32_t *tile_wrow = nullptr;
    uint4 *ztab = nullptr;              // draw tables (k_refresh_ztab)
    uint32_t *seg_aff = nullptr;        // implicit adjacency of table segments
    uint32_t *seg_wide = nullptr, *wide_exc = nullptr;   // wide quads of table segments (nsk_compile.h)
    uint8_t *sink = nullptr;            // scratch line for padding lanes' stores
    uint32_t *hub_desc = nullptr, *hub_adj = nullptr;   // entry-parallel hub streams
    uint32_t *ep_desc = nullptr, *ep_adj = nullptr;     // entry-parallel groups of general tiles
    uint32_t *bighub_pos = nullptr, *ep_wrow = nullptr, *ep_kstat = nullptr, *ep_win = nullptr, *ep_win_off = nullptr;
    double *ep_wt = nullptr;
    nsk::ZProgDev *zprogs = nullptr;
    bool values_regular = true;         // every value on the device lies in [0, cardinality): the
                                        // table kernels index with the neighbours' low bits
    bool chain_regular[2] = {true, true};   // ... per chain (var_value, var_value_evid)
    double compile_seconds = 0;
    uint32_t *dyn_tiles = nullptr, *rest_tiles = nullptr, *learn_rest_tiles = nullptr;
    int acc_copies = 1;                // copies of the global learning accumulators (one per XCD, or 1)
    int bins_xcd = 0;                  // SMALLW bins private to XCDs (close_sink)
    bool generic_uploaded = false;     // CSR-style arrays of the generic kernels are on the device
    double learn_cap = 0.5;            // per-class cap on visits * step of one weight (nsk_set_learn_cap)
    unsigned int *clip_count = nullptr;   // weight updates whose step was clipped (device counter)
    long long *part_G = nullptr;       // SMALLW: NSK_LEARN_BINS bins of partial sums per weight
    uint32_t *part_K = nullptr, *part_T = nullptr;
    bool smallw = false;
    // One-class-lagged learning (nsk_set_learn_lag, default on): class c samples with the weights as of
    // the end of class c - 2 while the update of class c - 1 runs beside it (block 0 of the class's table
    // launch) or behind it; weights, slot-program terms, draw tables and accumulators exist twice (set 0 =
    // the arrays above) and the classes alternate between them.
    bool learn_lag = true;
    double *w1 = nullptr, *prog_w1 = nullptr;
    uint4 *ztab1 = nullptr;
    long long *G1 = nullptr, *part_G1 = nullptr;
    uint32_t *K1 = nullptr, *T1 = nullptr, *part_K1 = nullptr, *part_T1 = nullptr;
    // state transfer (nsk_state_upload / nsk_state_download): values cross PCIe narrowed (1 or 4 bytes) in the
    // caller's order through pinned staging buffers; the permutation to layout order runs on the device
    void *xfer_host[2] = {nullptr, nullptr};       // pinned, nvar * vbytes each (one per chain)
    void *xfer_dev = nullptr;                      // nvar * vbytes
    int32_t *xfer_iid = nullptr;                   // internal id of every variable
    void *cnt_host = nullptr;                      // pinned, ncount * 4: the tally crosses PCIe as int32 when it fits
    int32_t *cnt_dev32 = nullptr;
    unsigned int *cnt_wide = nullptr;
    bool weights_dirty = true;      // prog_w must be rebuilt before the next fast-path launch
    bool weights_exposed = false;
    std::vector<double> w_stage;        // host staging of weight transfers when the table is in slot order (wmap)
    bool adj_wt_skip = false;       // learning reads weights directly: skip the shape-tile rows until the next inference   // the weight buffer was handed out: assume it changes between calls
    // boundary exchange (multi-GPU)
    int xworld = 0, xrank = 0;
    int64_t xslot = 0, xnsend = 0, xnrecv = 0;
    int32_t *x_send_vids = nullptr, *x_recv_vids = nullptr, *x_recv_slot = nullptr;
    void *x_send = nullptr, *x_recv = nullptr, *x_send_evid = nullptr, *x_recv_evid = nullptr;
    double *w_start = nullptr, *w_delta = nullptr;
    // peer-to-peer exchange (nsk_p2p_*): pairwise boundary lists, ONE fine-grained allocation per rank
    // (flags | received values of both chains, two parities | weight deltas; nsk_kernels_misc.h), the
    // peers' mappings of theirs, the exchange counter
    int pworld = 0, prank = 0;
    int64_t p_nsend = 0, p_nrecv = 0;
    std::vector<int64_t> p_soff, p_roff, p_dbase, p_dtotal;
    int32_t *p_send_iid = nullptr, *p_recv_iid = nullptr;
    void *p2p_base = nullptr;
    size_t p2p_bytes = 0;
    unsigned int *p2p_err = nullptr;
    void *p2p_peer_base[16] = {nullptr};
    bool p2p_peer_ipc[16] = {false};                // mapped with hipIpcOpenMemHandle (closed at destroy / re-import)
    unsigned int p2p_peer_mask = 0, p2p_tag = 0;
    unsigned long long p2p_timeout_ticks = 3000000000ull;      // 30 s of the 100 MHz wall clock (NSK_P2P_TIMEOUT_S)
    bool p2p_ready = false;
    // partial-factor aggregates this rank computes for its readers (nsk_pf_setup): value slots [nid, nid + npf)
    int64_t npf = 0;
    uint8_t *pf_op = nullptr;
    int32_t *pf_off = nullptr, *pf_mem = nullptr;
    // Fused boundary exchange of the table launches (nsk_api.hip p2p_fuse_plan, nsk_kernels_gibbs.h TabP2P): the
    // inference sweeps of a handle that lives in table segments read their ghosts from the receive block and
    // push their boundary values from inside the class launches -- no exchange kernels per sweep
    bool p2p_fused = false;                 // the handle qualifies (decided when the peers' buffers are imported)
    bool p2p_fused_now = false;             // ... and the running nsk_gibbs_sweeps_p2p call sweeps that way
    bool p2p_close_pending = false;         // the closing wait + unpack of the last fused sweep is not enqueued yet
    std::vector<int32_t> p2p_border_tiles;  // sorted: tiles (position >> 6) that own a value a peer reads or read a ghost;
                                            //   a tile's rank here is its row in the push map
    uint32_t *p2p_push_map = nullptr;       // [rows][64] reader << 28 | index in the reader's receive block; NSK_NO_STREAM
    uint32_t p2p_ghost_lo = 0, p2p_border_total = 0;   // first ghost id; border tiles one sweep of the current plans samples
    int p2p_first_phase = -1;               // the sweep's first class with table launches: its border tiles wait for the flags
    bool p2p_border_all = true;             // ... and that is every border tile (else the call takes the exchange kernels)
    std::vector<int32_t> p_send_host, p_recv_host;      // the send / receive lists (internal ids), host copies
    // native RCCL
    void *rccl_lib = nullptr, *rccl_comm = nullptr;
    long long *cnt_total = nullptr, *G = nullptr;
    uint32_t *K = nullptr, *T = nullptr;
    nsk::MTState *mt_np = nullptr, *mt_py = nullptr;
    // run state
    uint64_t seed = 0, sweep = 0;
    // Generator ids are positions in THIS handle's layout, so two shards of one graph would draw
    // the same uniforms at equal positions: Philox counter word 3 is the sweep index's high half
    // XOR this tag -- the first variable id the handle owns (disjoint shards => distinct tags;
    // 0 for a handle that owns the whole graph).
    uint32_t rng_tag = 0;
    int scan = NSK_SCAN_CHROMATIC;
    bool cnt_dirty = false;
    int64_t sweeps_done = 0;
    // profiling bracket
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t launches = 0, launches_at_begin = 0;
};

// Philox counter word 3 of every sweep kernel (see rng_tag)
static inline uint32_t nsk_sweep_hi(const nsk_graph *g) { return (uint32_t)(g->sweep >> 32) ^ g->rng_tag; }

template <typename VT>
static nsk::DevGraph<VT> view(nsk_graph *g) {
    nsk::DevGraph<VT> d;
    d.p_vid = g->p_vid; d.p_info = g->p_info; d.p_slot = g->p_slot; d.p_cnt = g->p_cnt;
    d.p_init = (const VT *)g->p_init;
    d.slot_off = g->slot_off; d.fidx = g->fidx;
    d.gstream = (const uint2 *)g->gstream; d.gs_off = g->gs_off;
    d.f_rec = (const uint4 *)g->f_rec; d.f_feat = g->f_feat;
    d.m_rec = (const int2 *)g->m_rec; d.v_card = g->v_card; d.iid_of_vid = g->iid_of_vid;
    d.w = g->w; d.w_fixed = g->w_fixed; d.logtab = g->logtab;
    d.val = (VT *)g->val; d.val_evid = (VT *)g->val_evid; d.cnt = g->cnt;
    d.G = g->G; d.K = g->K; d.T = g->T;
    d.adj = (const uint4 *)g->adj; d.tiles = (const uint4 *)g->tiles; d.tile_hdr = g->tile_hdr;
    d.prog_w = g->prog_w; d.adj_wt = g->adj_wt; d.tile_wrow = g->tile_wrow;
    d.part_G = g->part_G; d.part_K = g->part_K; d.part_T = g->part_T;
    d.nweight = (int32_t)g->c.nweight;
    d.packed_grad = g->c.packed_grad ? 1 : 0;
    d.grad_mul = 1ll << (32 - g->c.grad_shift);
    d.grad_inv = 1.0 / (double)d.grad_mul;
    d.acc_copies = g->acc_copies;
    d.bins_xcd = g->bins_xcd;
    d.cnt_pos = g->cnt_pos;
    d.ztab = g->ztab;
    d.sink = g->sink;
    d.hub_desc = (const uint4 *)g->hub_desc; d.hub_adj = g->hub_adj;
    d.ep_desc = (const uint4 *)g->ep_desc; d.ep_adj = g->ep_adj; d.bighub_pos = g->bighub_pos;
    d.ep_wrow = g->ep_wrow; d.ep_wt = g->ep_wt;
    d.ep_win = g->c.ep_win.empty() ? nullptr : g->ep_win; d.ep_win_off = g->ep_win_off;
    d.ep_kstat = g->c.ep_kstat.empty() ? nullptr : g->ep_kstat;
    d.seg_aff = (const uint4 *)g->seg_aff;
    d.seg_wide = g->seg_wide; d.wide_exc = (const uint2 *)g->wide_exc;
    d.w_direct = g->c.ndirect > 0 ? g->w_direct : nullptr;
    d.upd_step = 0.0; d.upd_reg_param = 0.0; d.upd_truncation = 1.0; d.upd_cap = 0.0; d.upd_regularization = 0; d.upd_a1 = 1.0;
    d.upd_clipped = g->clip_count;
    d.nvar = (int32_t)g->c.nvar;
    d.head_by_vid = (g->c.flags & NSK_FLAG_HEAD_BY_VID) ? 1 : 0;
    return d;
}

// Streams for the kernels of one colour: the tile kernels stay on the main stream; when there are
// tile kernels to overlap with, hubs go to side stream 0, the generic kernel to side stream 1 and
// the categorical general tiles to side stream 2.
struct ColourStreams {
    nsk_graph *g;
    bool forked[3] = {false, false, false};
    bool overlap;
    bool recorded = false;
    ColourStreams(nsk_graph *g_, bool overlap_) : g(g_), overlap(overlap_) {}
    // side streams must be requested before anything of the colour is put on the main stream
    hipStream_t side(int i) {
        if (!overlap) return g->stream;
        if (!recorded) { (void)hipEventRecord(g->ev_fork, g->stream); recorded = true; }
        if (!forked[i]) { (void)hipStreamWaitEvent(g->side[i], g->ev_fork, 0); forked[i] = true; }
        return g->side[i];
    }
    void join() {
        for (int i = 0; i < 3; i++)
            if (forked[i]) {
                (void)hipEventRecord(g->ev_join[i], g->side[i]);
                (void)hipStreamWaitEvent(g->stream, g->ev_join[i], 0);
                forked[i] = false;
            }
    }
};

// the fast path reads weights through prog_w (and the draw tables): rebuilt whenever weights may
// have changed (nsk_api.hip)
// Grid of a table-driven learning segment launch: resident -- the waves loop over their XCD's trips
// with the next trip's loads in flight (k_learn_seg_tab) -- and smaller when a block's flush of its
// LDS sums (a few atomics per weight) would rival its tile traffic
static inline int nsk_learn_tab_grid(int ntiles, int nweight, bool smallw) {
    const int blocks = (ntiles + 7) / 8;                            // 4 waves x 2 tiles
    const int trips = smallw ? std::max(1, (nweight * 16 * 16 + 14847) / 14848) : 1;
    // Six blocks per CU.  The kernel holds 7 waves per SIMD (112 scalar registers), so 2048 blocks are
    // not all resident, and whole blocks per CU beat fractions (measured on the 10M grid, per class with
    // the update launch: 1024 blocks 31.2 us, 1280 30.1, 1408 31.1, 1536 29.6, 1664 31.2, 1792 32.2,
    // 1920 32.5, 2048 31.4, 3072 31.3, 4096 31.7; the 1M grid is indifferent: 14.1-14.3)
    const char *cap_env = nsk::diag_env("NSK_LEARN_GRID_CAP");              // (diagnostic; read per launch so that tests can set it)
    // (SMALLW launches carry 8 service blocks in front, k_learn_seg_tab: they count towards the six per CU)
    const int cap = cap_env ? atoi(cap_env) : (smallw ? 1528 : 1536);
    return 8 * ((std::max(1, std::min(cap, (blocks + trips - 1) / trips)) + 7) / 8);     // whole rounds of XCDs
}
// Grid of a table-driven inference segment launch (k_gibbs_seg_tab) over `vtiles` virtual tiles (a multiple
// of 4): one wave per tile PAIR while that fits the resident grid, else the resident grid -- 6 blocks per CU,
// whole rounds of XCDs -- whose waves loop over QUADS (the kernel deals whole rounds of quads, then pairs).
// Six, not the seven the compiler's occupancy and the occupancy API report: the kernel holds 106 scalar
// registers and the hardware admits min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) blocks of 256 threads per CU
// (MI355X_MICROARCH.md); capping the registers for a seventh or eighth block was slower.  A grid beyond the resident one runs a tail round as long as
// the first: per 10M-grid class 1024 blocks 15.2 us, 1280 14.3, 1536 12.7, 1632 16.2, 1792 14.1, 2048 14.3
// (NSK_TAB_GRID_CAP); the 1M grid (1954 blocks of pairs) is indifferent (4.2 us).
static inline int nsk_tab_grid(int vtiles) {
    const int npairs = vtiles / 2;
    const int need = std::max(8, 8 * ((((npairs + 3) / 4) + 7) / 8));
    const char *cap_env = nsk::diag_env("NSK_TAB_GRID_CAP");                // (diagnostic)
    const int cap = cap_env ? std::max(8, std::abs(atoi(cap_env)) & ~7) : (need <= 2048 ? 2048 : 1536);
    return std::min(cap, need);        // (a launch that needs fewer blocks than 2048 runs in pairs: one trip per wave)
}
// Grid of a wide-quad table launch (k_gibbs_seg_tabw) over `vtiles` virtual tiles: a wave per quad while that fits the
// resident grid (NSK_TABW_PER_CU blocks per CU), else the resident grid, whole rounds of XCDs
#ifndef NSK_TABW_PER_CU
#define NSK_TABW_PER_CU 6
#endif
static inline int nsk_tabw_grid(int vtiles) {
    const int nquads = vtiles / 4, ntrips = nquads + 8;                     // (+ 8: every XCD's share rounds up)
    const int need = std::max(8, 8 * ((((ntrips + 3) / 4) + 7) / 8));
    const char *cap_env = nsk::diag_env("NSK_TABW_GRID_CAP");               // (diagnostic)
    const int cap = cap_env ? std::max(8, std::abs(atoi(cap_env)) & ~7) : 256 * NSK_TABW_PER_CU;
    return std::min(cap, need);
}
// The quads of a wide launch that are not wide ones (TabwCold.rest / TabwRest; the kernels' own tests of a quad, mirrored):
// each gets a workgroup in front of the grid.  More than NSK_TABW_REST_MAX of them: none listed, the waves sample them in
// line (NSK_DIAG=1 NSK_NO_TABW_REST=1: likewise).
static inline void nsk_tabw_rest_list(const nsk::Compiled &c, const nsk::SegTable &tab, int nch, uint32_t &nrest, uint32_t *out) {
    const int ST = NSK_WIDE_STRIDE(nch);
    std::vector<uint32_t> rest;
    for (int i = 0; i < tab.n; i++) {
        const nsk::SegEntry &en = tab.e[i];
        const int tend = i + 1 < NSK_SEG_MAX ? tab.e[i + 1].tile_start : tab.ntiles;
        const int lead = (int)(en.ntiles_lead >> 30), nt = (int)(en.ntiles_lead & 0x3FFFFFFFu);
        const int qs = en.tile_start >> 2, qin_lo = qs + (lead ? 1 : 0);
        const uint32_t qin_n = (uint32_t)(qs + ((lead + nt) >> 2) - qin_lo);
        const bool hasw = en.wide_off != NSK_NO_STREAM;
        for (int Q = qs; Q < (tend >> 2); Q++) {
            const bool flagged = hasw && c.seg_wide[(size_t)en.wide_off + (size_t)(Q - qs) * ST] != 0xFFFFFFFFu;
            if (flagged && (uint32_t)(Q - qin_lo) < qin_n) continue;     // a wide quad
            rest.push_back((uint32_t)Q | (flagged ? 0x80000000u : 0u));
        }
    }
    nrest = 0;
    if (rest.size() <= NSK_TABW_REST_MAX && !nsk::diag_env("NSK_NO_TABW_REST")) {
        nrest = (uint32_t)rest.size();
        for (size_t k = 0; k < rest.size(); k++) out[k] = rest[k];
    }
}
static inline int nsk_tabw_front_blocks(uint32_t nrest) { return (int)((nrest + 7u) & ~7u); }     // whole rounds of XCDs

// Grid of a wide-quad learning launch (k_learn_seg_tabw) over `vtiles` virtual tiles, whole rounds of XCDs (the service
// and front blocks come on top).  One MI355X, us per learning sweep (tools/sessions/r6_s26.sh): 10M grid 45.6 / 45.6 / 40.3 /
// 42.6 / 41.8 / 43.7 / 43.3 at 512 / 640 / 768 / 896 / 1024 / 1152 / 1280 blocks (45.5 tile by tile; 41.0 at 1024 and 43.0 at 1536 in
// r6_s25.sh); 40M grid 120.8 / 117.7 / 117.4 / 111.6 at 768 / 1024 / 1536 / 2048 (146.6 tile by tile).  (While the waves whose
// turn they were sampled the launch's few quads that are not wide in line, these figures jumped by a quarter from one
// grid size to the next -- whichever waves drew those quads ended the launch: r6_s16.sh.)  On the 4M grid the
// tile-by-tile kernel wins (24.3 against 27.5), hence NSK_WIDE_LEARN_MIN_QUADS.
#define NSK_WIDE_LEARN_MIN_QUADS 12000
static inline int nsk_learn_tabw_grid(int vtiles) {
    const int nquads = vtiles / 4 + 8;
    const int need = std::max(8, 8 * ((((nquads + 3) / 4) + 7) / 8));
    const char *cap_env = nsk::diag_env("NSK_LEARN_TABW_GRID_CAP");         // (diagnostic)
    const int cap = cap_env ? std::max(8, std::abs(atoi(cap_env)) & ~7) : (nquads >= 40000 ? 2048 : 1024);
    return std::min(cap, need);
}
static inline int nsk_learn_seg_grid(const nsk::Compiled::SegLaunch &sl, int nweight, bool smallw, bool use_tab) {
    const int ntiles = sl.tile_start[sl.n];
    if (sl.tab && use_tab) return nsk_learn_tab_grid(ntiles, nweight, smallw);
    return std::min(NSK_LEARN_SEG_BLOCKS, (ntiles + 3) / 4);
}

extern "C" int nsk_ensure_generic(nsk_graph *g);       // internal (not in the public header)
int nsk_ensure_lag_sets(nsk_graph *g);                 // second set of weights / tables / accumulators (nsk_api.hip)
#define NSK_GRAPH_SWEEPS 16
#define NSK_GRAPH_SWEEPS_BIG 64
// one peer-to-peer exchange on the library's stream; tag_base != null: a captured launch whose tag is
// the device counter + tag_off; learn: both chains + the weight deltas; part 0 = all of it, 1 = the
// pushes only, 2 = wait + unpack (+ the owner's half of the weight merge), 3 = the closing half of the
// weight merge (nsk_api.hip)
int nsk_p2p_enqueue(nsk_graph *g, const unsigned long long *tag_base, unsigned int tag_off, bool learn = false, int part = 0);
bool nsk_tables_only(const nsk_graph *g);           // every sampled variable lives in a table segment (nsk_api.hip)
void nsk_ensure_seg_plans(nsk_graph *g, int sample_evidence);    // nsk_gibbs.hip
int nsk_p2p_ghost_pack(nsk_graph *g);
int nsk_p2p_flush(nsk_graph *g);                   // enqueue the pending closing wait + unpack of a fused sweep sequence, if any
void nsk_p2p_fill(nsk_graph *g, nsk::TabP2P &px, const unsigned long long *tag_base, unsigned int tag, bool wait);   // kernel argument of a fused launch
void nsk_drop_sweep_graph(nsk_graph *g);            // the captured sweep sequence bakes exchange pointers: drop it when they change
int nsk_gibbs_run(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin, bool p2p);   // nsk_gibbs.hip
void nsk_refresh_prog_weights(nsk_graph *g, bool force = false);
void nsk_refresh_ztab(nsk_graph *g, int set = 0, hipStream_t st = nullptr);
int nsk_fold_position_tally(nsk_graph *g);
int nsk_unpack_tally(nsk_graph *g);
