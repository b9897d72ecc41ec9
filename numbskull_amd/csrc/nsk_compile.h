// nsk_compile.h -- host-side "graph compiler": validates a reference-layout factor graph,
// colours it and lays it out for the device (DESIGN.md "Data layout in HBM").
#pragma once

#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/numbskull_amd.h"

#define NSK_LEARN_SEG_LAUNCHES 4       // segment launches of the learning sweep per colour class
#define NSK_WIDE_STRIDE(nch) (4 * (nch) + 4)   // dwords of a quad descriptor in seg_wide
#define NSK_WIDE_MAXEXC 8          // exceptions a wide quad may carry
#define NSK_GEN_NULL 0x7FFFFFFu     // member id of an empty slot in a general tile (kind 6)
// (value windows of the entry-parallel groups: measured and NOT kept -- compiled in only with -DNSK_EP_WIN, see ep_win below)
#define NSK_EP_WIN_CHUNKS 512       // 16-byte chunks of a group's value window (8 KB per chain in LDS)
#define NSK_EP_WIN_BASE 0x7FF0000u  // member ids from here on (below NSK_GEN_NULL) are offsets into the window

namespace nsk {

// Static block partition of [0, n) over the host threads (NSK_COMPILE_THREADS, default: the hardware's, at
// most 64).  Every use writes disjoint outputs per index, so results do not depend on the thread count.
static inline int compile_threads() {
    static const int n = [] {
        const char *e = getenv("NSK_COMPILE_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(64, t));
    }();
    return n;
}
template <typename F>
static void parallel_for(int64_t n, F &&body, int64_t grain = 4096) {    // body(begin, end, thread index); >= grain items per thread
    const int T = (int)std::min<int64_t>(compile_threads(), std::max<int64_t>(1, n / grain));
    if (T <= 1) { body((int64_t)0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] { body(n * t / T, n * (t + 1) / T, t); });
    for (auto &x : th) x.join();
}

// Diagnostic switches (INTEGRATION.md "Diagnostic switches") change which layout / kernel family a
// graph compiles to -- and therefore its sample stream.  A product library must not pick those up
// from an inherited environment: they are read only when NSK_DIAG=1 is set as well.
inline const char *diag_env(const char *name) {
    const char *on = getenv("NSK_DIAG");
    if (!on || on[0] != '1') return nullptr;
    return getenv(name);
}

struct Compiled {
    // sizes
    int64_t nvar = 0, nweight = 0, nfactor = 0, nedge = 0, ncount = 0;
    int64_t npos = 0, nslot = 0;    // npos counts padding positions too
    int64_t nsampled = 0;           // variables this handle samples
    int vbytes = 1;                 // 1: int8 values, 4: int32 values
    int flags = 0;
    int64_t own_begin = 0, own_end = 0;
    // colouring
    std::vector<int32_t> color;         // [nvar], -1 = not sampled by this handle
    std::vector<int64_t> phase_start;   // [ncolors+1] positions (every colour starts on a multiple of 128)
    std::vector<int64_t> phase_end;     // [ncolors] end of the colour's positions (<= phase_start[k+1])
    // Inside a phase the "fast" variables (binary, symmetric boolean factors: inlined adjacency
    // streams) come first, the rest (generic CSR kernel) after: [phase_start, phase_fast_end) fast.
    std::vector<int64_t> phase_fast_end;   // [ncolors]
    // generic range of a phase: [phase_fast_end, phase_heavy_end) hubs (one wave per variable),
    // [phase_heavy_end, phase_start[k+1]) one lane per variable
    std::vector<int64_t> phase_heavy_end;  // [ncolors]
    std::vector<int64_t> phase_wb_base;    // [ncolors+1] first wave-block of each phase
    // One 16-byte descriptor per wave-block ("tile" = 64 consecutive fast positions):
    //   [0] offset of the tile's stream in 16-byte units   [1] words per lane (multiple of 4)
    //   [2] offset into tile_hdr of a uniform tile's slot program, 0xFFFFFFFF = per-lane headers
    //   [3] member slots of a uniform tile
    // Stream layout: chunk c of lane i (4 words = one 16-byte load) at 16*(off + 64*c + i).
    std::vector<uint32_t> tiles;           // [4*nwb]
    // shape tiles: first row of the tile in the materialised weight stream (adj_wt: one row of 64
    // doubles per entry, refreshed from the weights whenever they change); 0 for other tiles
    std::vector<uint32_t> tile_wrow;       // [nwb]
    int64_t nwrows = 0;
    std::vector<uint32_t> adj;             // inlined adjacency words (DESIGN.md "fast path")
    // entry-parallel hub streams: per hub position {offset, entries, M | cardinality << 8, 0} (entries == 0:
    // the hub takes the generic walk) and the lane-per-entry word rows
    std::vector<uint32_t> hub_desc, hub_adj;
    std::vector<int64_t> phase_hub_base;    // [ncolors+1] first descriptor of each colour's hub range
    // hubs with long lists in entry-parallel colours: positions, colour by colour (hub_desc[..][3] = 1)
    std::vector<uint32_t> bighub_pos;
    std::vector<int64_t> phase_bighub_base; // [ncolors+1]
    int64_t nhub_ep = 0;
    // A tile whose 64 lanes share one header sequence (same function, member count and weight per
    // entry) with at most 8 member slots is "uniform": its stream holds member words only and its
    // per-slot program (weight id, function code, first/last/ignore flags; nsk_compile.cpp) is kept
    // once in tile_hdr, padded to 8 words, for the scalar unit to read.
    std::vector<uint32_t> tile_hdr;
    // first positions of the tiles with per-lane headers, per phase (the learning sweep hands
    // them to the generic kernel): dyn_tiles[phase_dyn_base[k] .. phase_dyn_base[k+1])
    std::vector<uint32_t> dyn_tiles;
    std::vector<int64_t> phase_dyn_base;
    // Homogeneous segments: runs of >= NSK_SEG_MIN_TILES consecutive FULL uniform tiles with one
    // program, slot count, kind and evidence flag.  The inference sweep launches a straight-line
    // kernel per segment with the tile description in kernel arguments; the tiles outside any
    // segment (rest_tiles, per phase) go through the descriptor-driven kernel.
    struct Segment { int32_t phase; int64_t pos0; int32_t ntiles; uint32_t adj_off, prog, nslots, kind; int32_t ev;
                     int64_t ztab;         // first entry of the program's draw table, -1 = none
                     int64_t aff;          // first entry of the segment's tiles in seg_aff (-1: none)
                     int64_t wide = -1; }; // first dword of the segment's quad descriptors in seg_wide (-1: none)
    std::vector<Segment> segments;
    // Implicit adjacency of table segments: a tile whose every member slot holds position
    // (base of the slot) + lane -- the interior of a grid, where a class's neighbours are runs of the
    // other class -- needs no stream: seg_aff holds one uint4 of slot bases per tile and 16-byte chunk
    // (x = 0xFFFFFFFF: the tile is not of that form and reads its stream).  The member gathers of such
    // a tile are contiguous 64-byte reads.
    std::vector<uint32_t> seg_aff;
    // Wide quads of table segments (int8 values).  A QUAD = the four tiles at positions 256 m .. 256 m + 255.  When
    // all four belong to one segment and every member slot j of (almost) every position 256 m + o holds id
    // base_j + o, ONE lane can take four consecutive positions: lane l loads the dword at base_j + 4 l per slot
    // (four neighbour bytes at once, 256 bytes per wave-instruction instead of 64), packs four neighbourhoods with
    // three shifts, and stores four new values / four tally bytes as dwords.  seg_wide holds per quad of the segment
    // (quad i of a segment = absolute quad (pos0 >> 8) + i; NSK_WIDE_STRIDE(nch) dwords each):
    //   [0 .. 4 nch) slot bases (word 0 = 0xFFFFFFFF: not a wide quad -- it is sampled tile by tile),
    //   [4 nch] first exception in wide_exc, [4 nch + 1] number of exceptions, [4 nch + 2] mask of the real member
    //   slots (a slot beyond the program's, or one that names the always-zero id, repeats base 0 and is masked out).
    // wide_exc: pairs {offset in the quad | slot << 8, true member id} -- the few positions whose member is NOT at
    // base + offset (the end cell of a grid row, whose neighbour lives in the border class); the owning lane
    // patches that one bit after the wide loads.  A wide quad draws from the WIDE generator scheme (nsk_device.h).
    std::vector<uint32_t> seg_wide, wide_exc;
    int64_t ntab_quads = 0, nwide_quads = 0;    // quads of table segments / the wide ones among them
    // Draw tables (DESIGN.md "draw tables"): a uniform program whose lanes read binary members only
    // has 2^nslots possible neighbourhoods; per neighbourhood the draw threshold and the per-slot
    // satisfied bits are tabulated by k_refresh_ztab whenever weights change.
    struct ZProg { uint32_t prog, nslots, off, pad; };
    std::vector<ZProg> zprogs;
    int64_t nztab = 0;                      // entries (16 bytes each)
    bool values_regular = true;             // every initial value lies in [0, cardinality)
    bool has_ufo = false;                   // a reachable factor is UFO: values index its member list
    double grad_bound = 0.0;                // bound on |gradient sum| of one weight in one colour class
    int grad_shift = 0;                     // gradients accumulate as Q(31+s).(32-s) fixed point: s > 0 when
                                            // the bound reaches 2^30 (Q31.32 would overflow)
    std::vector<uint32_t> rest_tiles;       // tile indices relative to the phase's first tile
    // learning: the largest (kind, chunks) groups of a colour's segments run as segment launches of
    // their own (at most NSK_LEARN_SEG_LAUNCHES per colour); every other non-general tile is on
    // the colour's learn_rest list
    struct SegLaunch { int32_t phase, kind, nch, n, tab; int32_t tile_start[9]; int32_t pos0[8];
                       uint32_t adj_off[8], prog[8], zoff[8], zmask[8], aff[8]; int32_t ev[8];
                       int64_t wide[8]; };    // first dword of the segment's quad descriptors in seg_wide (-1: none)
    std::vector<SegLaunch> learn_seg;
    std::vector<uint32_t> learn_rest_tiles;
    std::vector<int64_t> phase_learn_rest_base;   // [ncolors+1]
    std::vector<int64_t> phase_rest_base;   // [ncolors+1]
    // general tiles (kind 6) are the last tiles of a colour's tile range: first one, relative
    std::vector<int64_t> phase_gen_tile;    // [ncolors]
    bool packed_grad = false;               // integer gradients, bounded visit counts (GradSink::packed)
    std::vector<int64_t> phase_gen_bin_tile;   // [ncolors] first of the all-binary general tiles (they come last)
    // Entry-parallel layout of a colour's general tiles (DESIGN.md "entry-parallel groups"): a GROUP is
    // four consecutive general tiles (256 positions = one workgroup).  Its list entries -- the very
    // words of the general tiles plus `ordinal << 27` in the weight word (position in the variable's
    // list order) and `position in the group << 23 | weight fixed << 31` in the descriptor word -- are
    // sorted into 8 classes (member count M = 0..3 of the entries with ordinal < 8, then the same for
    // ordinals 8..15) and cut into ROWS of 64 entries, one entry per lane: a row is (2 + M) sub-rows
    // of 64 words in ep_adj -- the (weight, descriptor) pairs interleaved in the first two, then one
    // sub-row per member slot.  No padding to the widest lane of a tile.
    // ep_desc[group] = {first sub-row, rows of classes 0-3 (8 bits each), most entries of a variable |
    // largest cardinality << 8, rows of classes 4-7}.
    // ep_wrow[group] = first row of the group in the materialised weight rows (one double per entry
    // slot, row-major: the inference kernels read an entry's weight next to its words; refreshed
    // whenever weights change); ep_wrow[ngroups] = rows in total
    std::vector<uint32_t> ep_desc, ep_adj, ep_wrow;
    // Value windows of the groups (int8 values only; -DNSK_EP_WIN builds, NSK_DIAG=1 NSK_NO_EP_WIN=1 switches them
    // off again): the members of a group's entries lie in a few short runs of every other colour.  ep_win[
    // ep_win_off[g] .. ep_win_off[g + 1]) lists the 16-byte chunks (internal id >> 4) that hold them, ascending, at
    // most NSK_EP_WIN_CHUNKS; the kernels copy them into LDS once per group (both chains) and a member word whose
    // id field is NSK_EP_WIN_BASE + o reads byte o of that copy instead of gathering from the value array.
    // MEASURED, round 5 (tools/sessions/r5_s02.sh; 91 % of the LR graphs' members inside a 512-chunk window): no
    // fewer bytes fetched (inference 50M: FETCH_SIZE 6.93e5 against 6.95e5 -- the value lines were never what
    // missed; what the value gathers cost the learning launch is L2 room for the 8 MB weight table) and slower
    // launches (5M LR: inference 50.8 -> 52.0 us per class, learning 114.5 -> 120.6; 50M inference 468 -> 497):
    // the 8 / 16 KB of LDS per workgroup cost more residency than the LDS reads save.  Not in the default build.
    std::vector<uint32_t> ep_win, ep_win_off;
    // Structural visit counts of the entry-parallel groups (learning): an entry of a dataType-0 variable
    // is visited by sample_and_sgd in EVERY sweep its variable takes part in (learning.py:76-95: one
    // list), so its contribution to the visit count K of its weight in its colour class is known here:
    // ep_kstat[(2 k + o) * nweight + w] = entries of colour k with weight w (not fixed) whose variable
    // is evidence (o = 0: takes part always) / is not (o = 1: only with learn_non_evidence).  The
    // gradient pass then skips the accumulator update of such an entry when its gradient is 0 -- the
    // common case -- and the weight update adds the structural count.  Empty: not used.
    std::vector<uint32_t> ep_kstat;
    std::vector<int64_t> phase_ep_base;        // [ncolors+1] first group of each colour
    std::vector<uint8_t> phase_ep;             // [ncolors] 1: the colour's general tiles are laid out as groups
    std::vector<int32_t> phase_ep_emax;        // [ncolors] most entries of one of its variables
    int64_t nfast = 0;
    // per position
    std::vector<int32_t> p_vid, p_slot, p_cnt;
    std::vector<uint32_t> p_info;
    std::vector<int32_t> p_init;        // narrowed to vbytes at upload
    // inverted index
    std::vector<int32_t> slot_off, fidx;
    // inline generic stream (one-lane-per-variable generic kernels): see nsk_device.h gstream
    std::vector<uint32_t> gstream;      // pairs of uint32 = 8-byte units
    std::vector<uint32_t> gs_off;       // [nslot]
    // per factor / edge / variable
    std::vector<uint32_t> f_rec;        // [4*nfactor] {arity << 8 | function+1, ftv_offset, weightId, 0}
    std::vector<double> f_feat;
    std::vector<int32_t> m_rec;         // [2*nedge] {variable id, dense_equal_to}
    std::vector<int32_t> v_card, v_pos;
    // internal numbering: iid[v] = position of a sampled variable, npos.. for the others; nid ids
    // (+ one id past them that belongs to no variable and always holds 0: zero_id)
    std::vector<int32_t> iid, v_card_i;
    int64_t nid = 0, zero_id = 0;
    bool literal_heads = false;         // a reachable factor reads its head at the literal edge index
    std::vector<int64_t> cstart;        // [nvar+1]
    // weights
    std::vector<double> w_init;
    std::vector<uint8_t> w_fixed;
    // Weights with ONE factor ("direct"): the members of a factor sit in different colour classes, so such a
    // weight has at most one visit per class and the learning kernels apply its update in place at that
    // visit -- no accumulator, no pass of the update launch over it.  Enabled when at least half of the
    // weights qualify (one weight per factor: feature-weighted graphs); bit w of w_direct; multi_wids = the
    // other weights, the only ones the update launch then walks.  Weights that a uniform tile's program
    // names stay with the accumulators (those kernels sum per tile, not per visit).
    std::vector<uint32_t> w_direct;
    std::vector<int64_t> repeated_factors;     // factors some variable lists twice in one list (never direct: two visits per class)
    std::vector<int32_t> multi_wids;
    int64_t ndirect = 0;
    // Internal numbering of the direct weights (whole-graph handles): with one weight per factor the table
    // is far larger than the caches and a variable's weights, numbered by the caller's factor order, sit
    // one per cache line.  The direct weights are renumbered AMONG THEMSELVES (their id set is kept: w_direct,
    // w_fixed, multi_wids and every program naming another weight stay as they are) in the order the layout
    // first meets them, so the entries of a group read and update neighbouring slots.  wmap: caller's id ->
    // slot in the device table (w_init is in slot order), wuser: the inverse; empty = identity.
    std::vector<int32_t> wmap, wuser;
    std::vector<double> logtab;
    // multi-GPU: variables outside the owned range that the sampled variables read (sorted)
    std::vector<int32_t> ghost_needs;
    // initial values by variable id
    std::vector<int32_t> v_init;
    // algorithmic traffic, bytes per sweep over the sampled variables
    double alg_bytes_inference = 0, alg_bytes_learning = 0;
    // bytes one sweep must move in THIS layout (tile words, position arrays, distinct values read
    // per colour class, stores, tallies): what roofline fractions are computed from
    double layout_bytes_inference = 0, layout_bytes_learning = 0;
};

// returns NSK_OK or an NSK_E_* code with `err` filled
int compile_graph(const nsk_graph_desc *d, Compiled &out, std::string &err);

// true for function ids of inference.py:74-143
bool known_function(int fn);

}  // namespace nsk
