// nsk_compile.h -- host-side "graph compiler": validates a reference-layout factor graph,
// colours it and lays it out for the device (DESIGN.md "Data layout in HBM").
#pragma once

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"

namespace nsk {

struct Compiled {
    // sizes
    int64_t nvar = 0, nweight = 0, nfactor = 0, nedge = 0, ncount = 0;
    int64_t npos = 0, nslot = 0;
    int vbytes = 1;                 // 1: int8 values, 4: int32 values
    int flags = 0;
    int64_t own_begin = 0, own_end = 0;
    // colouring
    std::vector<int32_t> color;         // [nvar], -1 = not sampled by this handle
    std::vector<int64_t> phase_start;   // [ncolors+1] positions
    // Inside a phase the "fast" variables (binary, symmetric boolean factors: inlined adjacency
    // streams) come first, the rest (generic CSR kernel) after: [phase_start, phase_fast_end) fast.
    std::vector<int64_t> phase_fast_end;   // [ncolors]
    std::vector<int64_t> phase_wb_base;    // [ncolors+1] first wave-block of each phase
    std::vector<uint32_t> wb_off;          // [nwb] word offset of a wave-block's stream
    std::vector<uint32_t> wb_len;          // [nwb] words per lane (column-major: word j of lane i
                                           //       at wb_off + 64*j + i)
    std::vector<uint32_t> adj;             // inlined adjacency words (DESIGN.md "fast path")
    // A tile whose 64 lanes share one header sequence (same function, member count and weight per
    // entry) keeps the headers once, in tile_hdr, and its stream holds member words only.
    std::vector<uint32_t> wb_hdr;          // [nwb] offset into tile_hdr, 0xFFFFFFFF = per-lane headers
    std::vector<uint32_t> wb_nent;         // [nwb] entries of a uniform tile
    std::vector<uint32_t> tile_hdr;
    int64_t nfast = 0;
    // per position
    std::vector<int32_t> p_vid, p_slot, p_cnt;
    std::vector<uint32_t> p_info;
    std::vector<int32_t> p_init;        // narrowed to vbytes at upload
    // inverted index
    std::vector<int32_t> slot_off, fidx;
    // per factor / edge / variable
    std::vector<uint32_t> f_head;
    std::vector<int32_t> f_off, f_wid;
    std::vector<double> f_feat;
    std::vector<int32_t> m_vid, m_deo;
    std::vector<int32_t> v_card, v_pos;
    std::vector<int64_t> cstart;        // [nvar+1]
    // weights
    std::vector<double> w_init;
    std::vector<uint8_t> w_fixed;
    std::vector<double> logtab;
    // initial values by variable id
    std::vector<int32_t> v_init;
    // algorithmic traffic, bytes per sweep over the sampled variables
    double alg_bytes_inference = 0, alg_bytes_learning = 0;
};

// returns NSK_OK or an NSK_E_* code with `err` filled
int compile_graph(const nsk_graph_desc *d, Compiled &out, std::string &err);

// true for function ids of inference.py:74-143
bool known_function(int fn);

}  // namespace nsk
