// nsk_compile.cpp -- validate + colour + lay out a factor graph for the device.
//
// Input: the arrays a reference FactorGraph is built from (factorgraph.py:30-37) in their packed
// numpy layouts.  Output: the SoA device layout of DESIGN.md.  Nothing here runs per sweep.
#include "nsk_compile.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <unordered_map>

namespace nsk {


// a variable whose factor lists hold at least this many entries in total is sampled by a whole wave
static const int64_t NSK_HEAVY_LIST = 32;
// ... and so is every generic-path variable of a colour class that has at most this many of them
static const int64_t NSK_FEW_GENERIC = 32768;
// stream words per lane of a shape tile at most (role program in TileShape::key, 32 words).  A lane walks its
// words chunk by chunk, every chunk a load and then its gathers: longer lists belong to the entry-parallel
// groups.  Measured on the 4M-variable weighted boolean graph (tools/sessions/history/r4_s24.sh, r4_s25.sh; learning /
// inference, updates/s): 16 words 3.68e9 / 1.28e10, 20 words 2.93e9 / 1.18e10, 24 words 1.72e9 / 1.11e10,
// 32 words 2.2e8 / 5.3e9.
static const int64_t NSK_SHAPE_WORDS = 16;
// ... and the same limit for variables the entry-parallel groups can take.  Once the single-factor weights had
// slots in layout order (nsk_compile.h wmap) the groups overtook the shape tiles at every list length; same
// graph, every rest tile with a wave of its own (tools/sessions/history/r4_s31.sh .. r4_s33.sh), learning / inference:
// 20 words 4.20e9 / 1.20e10, 16 words 4.78e9 / 1.30e10, 12 words 5.25e9 / 1.56e10, 10 words 5.61e9 / 1.60e10,
// 8 words 6.13e9 / 1.60e10, 6 words 6.35e9 / 1.59e10, 4 words 6.56e9 / 1.60e10.
static const int64_t NSK_SHAPE_WORDS_EP = 4;
// a member slot of a shape tile that a lane does not have (its entry has fewer members than the tile's layout)
static const uint32_t NSK_SHAPE_NULL = 0xFFFFFFFFu;

bool known_function(int fn) {
    switch (fn) {
    case -1: case 0: case 1: case 2: case 3: case 4: case 7: case 8: case 9:
    case 12: case 13: case 14: case 15: case 16: case 17:
    case 18: case 19: case 20: case 21: case 22: case 23: case 24: case 25: case 26:
    case 30:
        return true;
    default:
        return false;
    }
}

static bool is_cat_function(int fn) { return fn == 12 || (fn >= 14 && fn <= 17); }
static bool literal_head_function(int fn) { return fn == 13 || fn == 16 || fn == 17; }

static std::string fmt(const char *f, long long a = 0, long long b = 0, long long c = 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), f, a, b, c);
    return std::string(buf);
}

// Direct weights (nsk_compile.h w_direct): the weights with one factor, when at least half of all weights are of
// that kind.  nwb = number of tiles (weights a uniform tile's program names stay with the accumulators).
static void find_direct_weights(const nsk_graph_desc *d, Compiled &c, int64_t nwb, bool verbose) {
    const int64_t nw = c.nweight, nfac = c.nfactor;
    c.w_direct.clear(); c.multi_wids.clear(); c.ndirect = 0;
    if (nw > 256 && !diag_env("NSK_NO_DIRECT")) {
        std::vector<uint8_t> nfac_of((size_t)nw, 0);                 // factors per weight, saturating at 2
        for (int64_t f = 0; f < nfac; f++) {
            const int64_t wid = d->factor[f].weightId;
            if (wid >= 0 && wid < nw && nfac_of[(size_t)wid] < 2) nfac_of[(size_t)wid]++;
        }
        for (int64_t f : c.repeated_factors) {                       // a factor listed twice by one variable: two visits per class
            const int64_t wid = d->factor[f].weightId;
            if (wid >= 0 && wid < nw) nfac_of[(size_t)wid] = 2;
        }
        for (int64_t t = 0; t < nwb; t++) {                          // weights named by uniform tiles' programs
            const uint32_t *td = &c.tiles[4 * t];
            if (td[2] == 0xFFFFFFFFu || ((td[3] >> 8) & 7u) >= 6u) continue;
            for (uint32_t j = 0; j < (td[3] & 0xFFu); j++) {
                const uint32_t wid = c.tile_hdr[td[2] + j] & 0xFFFFFFu;
                if ((int64_t)wid < nw) nfac_of[wid] = 2;
            }
        }
        int64_t nd = 0;
        for (int64_t w = 0; w < nw; w++) nd += (nfac_of[(size_t)w] == 1 && !c.w_fixed[(size_t)w]) ? 1 : 0;
        if (2 * nd >= nw) {
            c.w_direct.assign((size_t)(nw + 31) / 32, 0u);
            for (int64_t w = 0; w < nw; w++) {
                if (nfac_of[(size_t)w] == 1 && !c.w_fixed[(size_t)w]) c.w_direct[(size_t)w >> 5] |= 1u << (w & 31);
                else c.multi_wids.push_back((int32_t)w);
            }
            c.ndirect = nd;
        }
        if (verbose) fprintf(stderr, "[nsk] weights with one factor %lld of %lld: %s\n", (long long)nd, (long long)nw,
                             c.ndirect ? "updated in place" : "too few, accumulators for all");
    }
}

// Internal numbering of the direct weights (nsk_compile.h wmap): the order in which the layout's positions, each
// walking its lists, first meet them; a weight no position names keeps the tail.  Returns false (and leaves the
// caller's numbering) on a handle that samples a range of a larger graph: the ranks of a distributed run add their
// weight tables element by element.
static bool number_direct_weights(const nsk_graph_desc *d, Compiled &c) {
    const int64_t nw = c.nweight, nvar = c.nvar, nfac = c.nfactor;
    c.wmap.clear(); c.wuser.clear();
    if (!(c.ndirect && c.own_begin == 0 && c.own_end == nvar && !(d->flags & NSK_FLAG_PARTITION) && !diag_env("NSK_NO_WORDER")))
        return false;
    auto is_direct = [&](int64_t w) { return (c.w_direct[(size_t)w >> 5] >> (w & 31)) & 1u; };
    std::vector<uint32_t> seen((size_t)(nw + 31) / 32, 0u);
    // (bands of 2^24 ids are numbered separately: a slot then has the bits of the id it replaces, and the
    // 24-bit weight field of the uniform-tile words holds whatever held before)
    std::vector<std::vector<int32_t>> order((size_t)((nw - 1) >> 24) + 1);
    for (int64_t p = 0; p < (int64_t)c.p_vid.size(); p++) {
        const int64_t v = c.p_vid[p];
        if (v < 0) continue;
        const nsk_variable &var = d->variable[v];
        const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
        for (int64_t k = 0; k < nslots; k++) {
            const nsk_vtf &vt = d->vmap[var.vtf_offset + k];
            for (int64_t j = 0; j < vt.factor_index_length; j++) {
                const int64_t w = d->factor[d->factor_index[vt.factor_index_offset + j]].weightId;
                if (w < 0 || w >= nw || !is_direct(w) || ((seen[(size_t)w >> 5] >> (w & 31)) & 1u)) continue;
                seen[(size_t)w >> 5] |= 1u << (w & 31);
                order[(size_t)w >> 24].push_back((int32_t)w);
            }
        }
    }
    for (int64_t w = 0; w < nw; w++)
        if (is_direct(w) && !((seen[(size_t)w >> 5] >> (w & 31)) & 1u)) order[(size_t)w >> 24].push_back((int32_t)w);
    c.wmap.resize((size_t)nw); c.wuser.resize((size_t)nw);
    std::vector<size_t> taken(order.size(), 0);
    for (int64_t w = 0; w < nw; w++) {
        if (!is_direct(w)) { c.wmap[(size_t)w] = (int32_t)w; c.wuser[(size_t)w] = (int32_t)w; continue; }
        const int32_t met = order[(size_t)w >> 24][taken[(size_t)w >> 24]++];
        c.wmap[(size_t)met] = (int32_t)w;               // the k-th weight of the band met takes its k-th direct slot
        c.wuser[(size_t)w] = met;
    }
    for (int64_t w = 0; w < nw; w++) c.w_init[(size_t)w] = d->weight[c.wuser[(size_t)w]].initialValue;
    for (int64_t f = 0; f < nfac; f++) {
        const int64_t w = d->factor[f].weightId;
        if (w >= 0 && w < nw) c.f_rec[4 * f + 2] = (uint32_t)c.wmap[(size_t)w];
    }
    return true;
}

// Gradient format of the learning accumulators.  Integer gradients?  (p1 - p0) * featureValue is an integer of
// magnitude <= 2 when featureValue is -1, 0 or 1 and no function returns counts or logarithms; visits per weight and
// class are bounded by the weight's member edges: the 32 fraction bits of G then carry the visit count (packed_grad).
// And the fixed-point range (grad_bound, grad_shift).
static void choose_gradient_format(const nsk_graph_desc *d, Compiled &c) {
    const int64_t nw = c.nweight, nfac = c.nfactor;
    bool ok = true;
    std::vector<int64_t> edges_of((size_t)nw, 0);
    for (int64_t f = 0; f < nfac && ok; f++) {
        const nsk_factor &fa = d->factor[f];
        const int fn = fa.factorFunction;
        if (!(fa.featureValue == 1.0 || fa.featureValue == 0.0 || fa.featureValue == -1.0)) ok = false;
        if (fn == 7 || fn == 8 || fn == 30) ok = false;            // LINEAR, RATIO, UFO
        if (fa.weightId >= 0 && fa.weightId < nw) edges_of[fa.weightId] += std::max<int64_t>(fa.arity, 1);
    }
    for (int64_t i = 0; i < nw && ok; i++) if (edges_of[i] >= ((int64_t)1 << 28)) ok = false;
    c.packed_grad = ok && !diag_env("NSK_NO_PACKED");
    // Q31.32 range: a class's gradient sum for weight w is at most sum over its factors of
    // |featureValue| * (largest |value difference| of the function) * (member edges)
    std::vector<double> gbound((size_t)nw, 0.0);
    for (int64_t f = 0; f < nfac; f++) {
        const nsk_factor &fa = d->factor[f];
        if (fa.weightId < 0 || fa.weightId >= nw) continue;
        const double ar = (double)std::max<int64_t>(fa.arity, 1);
        const int fn = fa.factorFunction;
        const double span = fn == 7 ? ar : fn == 8 ? std::log(ar + 1.0) : fn == 30 ? 1e6 : 2.0;
        gbound[fa.weightId] += std::fabs(fa.featureValue) * span * ar;
    }
    c.grad_bound = 0.0;
    for (int64_t i = 0; i < nw; i++) c.grad_bound = std::max(c.grad_bound, gbound[i]);
    // Q31.32 holds sums below 2^31; a larger bound trades fraction bits for range (the reference
    // sums float64 gradients, learning.py:109): Q(31+s).(32-s), gradients below 2^-(33-s) vanish
    c.grad_shift = 0;
    while (c.grad_shift < 32 && c.grad_bound >= 1073741824.0 * std::ldexp(1.0, c.grad_shift)) c.grad_shift++;
    if (c.grad_shift > 0) c.packed_grad = false;       // the fraction bits are no longer free for visit counts
}

// Validation of every factor reachable from a sampled variable (errors as the reference raises them: SURVEY.md
// section 8b); sets c.has_ufo / c.literal_heads, returns the largest arity of a RATIO factor in max_ratio_arity.
static int validate_reachable(const nsk_graph_desc *d, Compiled &c, const std::vector<uint8_t> &sampled, bool head_by_vid,
                              int64_t &max_ratio_arity_out, std::string &err) {
    const int64_t nvar = d->nvar, nfac = d->nfactor, nedge = d->nedge, nw = d->nweight;
    const int64_t nfi = d->nfactor_index;
    // ---- validate every factor reachable from a sampled variable ------------------------------
    // Two parallel phases over index blocks: (1) every sampled variable's lists -- bounds, factor ids -- mark
    // the factors they reach; (2) every reached factor is checked.  Each thread keeps the first error of its
    // block (lowest variable / factor index); the lowest block's error is reported, list errors first, so
    // the message does not depend on the thread count.
    std::vector<uint8_t> checked(nfac, 0);
    int64_t max_ratio_arity = 0;
    struct Issue { int rc = NSK_OK; std::string msg; bool ufo = false, literal = false; int64_t ratio = 0; };
    auto check_factor = [&](int64_t f, Issue &is) -> int {
        const nsk_factor &fa = d->factor[f];
        const int fn = fa.factorFunction;
        if (!known_function(fn)) {
            is.msg = fmt("Factor function %lld (used in factor %lld) is not implemented.", fn, f);
            return NSK_E_FACTOR_FUNC;
        }
        if (fa.weightId < 0 || fa.weightId >= nw) {      // potential() reads it even for NOOP
            is.msg = fmt("factor %lld: weightId %lld outside weights", f, fa.weightId);
            return NSK_E_INDEX;
        }
        if (fn == -1) return NSK_OK;
        const int64_t s = fa.ftv_offset, e = fa.ftv_offset + fa.arity;
        if (fa.arity < 0 || s < 0 || e > nedge) {
            is.msg = fmt("factor %lld: members [%lld, %lld) outside fmap", f, s, e);
            return NSK_E_INDEX;
        }
        int64_t need = 0;       // member positions the function reads regardless of arity
        switch (fn) {
        case 3: need = 1; break;
        case 0: case 7: case 8: case 9: case 13: case 16: case 17:
            if (fa.arity < 1) { is.msg = fmt("factor %lld: function %lld needs arity >= 1", f, fn); return NSK_E_INDEX; }
            break;
        case 18: case 19: case 20: case 30: need = 1; break;
        case 21: case 22: case 25: case 26: need = 2; break;
        case 23: case 24: need = 3; break;
        default: break;
        }
        const int64_t last = std::max(e, s + need);
        if (s + need > nedge) {
            is.msg = fmt("factor %lld: function %lld reads member %lld beyond fmap", f, fn, s + need - 1);
            return NSK_E_INDEX;
        }
        for (int64_t l = s; l < last; l++) {
            if (d->fmap[l].vid < 0 || d->fmap[l].vid >= nvar) {
                is.msg = fmt("factor %lld: member variable %lld outside variables", f, d->fmap[l].vid);
                return NSK_E_INDEX;
            }
        }
        if (fn == 30) is.ufo = true;
        if (fn == 30) {   // UFO reads member (value of first member) - 1
            int64_t reach = s + d->variable[d->fmap[s].vid].cardinality - 2;
            if (reach >= nedge) { is.msg = fmt("factor %lld: UFO member index beyond fmap", f); return NSK_E_INDEX; }
            for (int64_t l = s; l <= reach; l++)
                if (d->fmap[l].vid < 0 || d->fmap[l].vid >= nvar) {
                    is.msg = fmt("factor %lld: member variable outside variables", f);
                    return NSK_E_INDEX;
                }
        }
        if (literal_head_function(fn) && !head_by_vid) is.literal = true;
        if (literal_head_function(fn) && !head_by_vid && e - 1 >= nvar) {
            is.msg = fmt("factor %lld: the reference reads var_value[%lld] for the head of function %lld "
                         "(inference.py:243,277,292), outside the variable array; pass NSK_FLAG_HEAD_BY_VID "
                         "for the fmap[l].vid lookup", f, e - 1, fn);
            return NSK_E_INDEX;
        }
        if (fn == 8) is.ratio = std::max(is.ratio, fa.arity);
        return NSK_OK;
    };
    {
        std::vector<Issue> issues((size_t)compile_threads());
        parallel_for(nvar, [&](int64_t vb0, int64_t vb1, int t) {
            Issue &is = issues[(size_t)t];
            std::vector<int64_t> sorted_list;
            for (int64_t v = vb0; v < vb1 && !is.rc; v++) {
                if (!sampled[v]) continue;
                const nsk_variable &var = d->variable[v];
                const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
                for (int64_t k = 0; k < nslots && !is.rc; k++) {
                    const nsk_vtf &vt = d->vmap[var.vtf_offset + k];
                    if (vt.factor_index_length < 0 || vt.factor_index_offset < 0 ||
                        vt.factor_index_offset + vt.factor_index_length > nfi) {
                        is.msg = fmt("variable %lld: factor list outside factor_index", v);
                        is.rc = NSK_E_INDEX;
                        break;
                    }
                    // a factor twice in ONE list (compute_var_map never produces that, dataloading.py:68-81; a
                    // caller of the C-ABI may, and need not hand in sorted lists): its weight is then visited twice by
                    // one variable in one class and must not be updated in place at "its one visit"
                    // (find_direct_weights).  Bit 1 of checked[f] marks it: found on a sorted copy of the list when
                    // the list is not ascending already; the marks are atomic ORs (several threads reach one factor).
                    const int64_t *fl = d->factor_index + vt.factor_index_offset;
                    bool ascending = true;
                    for (int64_t j = 0; j < vt.factor_index_length; j++) {
                        const int64_t f = fl[j];
                        if (f < 0 || f >= nfac) {
                            is.msg = fmt("variable %lld: factor id %lld outside factors", v, f);
                            is.rc = NSK_E_INDEX;
                            break;
                        }
                        if (j > 0 && fl[j - 1] > f) ascending = false;
                    }
                    if (is.rc) break;
                    if (ascending) {
                        for (int64_t j = 0; j < vt.factor_index_length; j++)
                            __atomic_fetch_or(&checked[fl[j]], (uint8_t)((j > 0 && fl[j - 1] == fl[j]) ? 3 : 1), __ATOMIC_RELAXED);
                    } else {
                        sorted_list.assign(fl, fl + vt.factor_index_length);
                        std::sort(sorted_list.begin(), sorted_list.end());
                        for (size_t j = 0; j < sorted_list.size(); j++)
                            __atomic_fetch_or(&checked[sorted_list[j]], (uint8_t)((j > 0 && sorted_list[j - 1] == sorted_list[j]) ? 3 : 1), __ATOMIC_RELAXED);
                    }
                }
            }
        });
        for (const Issue &is : issues) if (is.rc) { err = is.msg; return is.rc; }      // blocks are in index order
        for (Issue &is : issues) is = Issue();
        parallel_for(nfac, [&](int64_t fb0, int64_t fb1, int t) {
            Issue &is = issues[(size_t)t];
            for (int64_t f = fb0; f < fb1 && !is.rc; f++)
                if (checked[f]) is.rc = check_factor(f, is);
        });
        c.repeated_factors.clear();
        for (int64_t f = 0; f < nfac; f++) if (checked[f] & 2) c.repeated_factors.push_back(f);
        for (const Issue &is : issues) {
            if (is.rc) { err = is.msg; return is.rc; }
            c.has_ufo = c.has_ufo || is.ufo;
            c.literal_heads = c.literal_heads || is.literal;
            max_ratio_arity = std::max(max_ratio_arity, is.ratio);
        }
    }
    max_ratio_arity_out = max_ratio_arity;
    return NSK_OK;
}

// Colouring of the sampled variables: no two variables of a colour may read each other.  Greedy first fit in id
// order, a symmetry check of the reads (repaired with reverse lists when a raw index is asymmetric), iterated greedy
// (class by class, the classes of a pass over the host threads) and a balancing pass.  for_each_read(v, fn) calls
// fn(b) for every variable b that v reads; lap(name) closes a timed stage.  Returns the number of colours.
template <typename ReadFn, typename LapFn>
static int32_t colour_sampled(Compiled &c, const std::vector<uint8_t> &sampled, ReadFn &&for_each_read, LapFn &&lap) {
    const int64_t nvar = c.nvar;
    c.color.assign(nvar, -1);
    std::vector<int64_t> stamp(1, -1), load;
    int32_t ncolors = 0;
    // greedy first fit in id order, then a balancing pass (below)
    auto pick = [&](int64_t v) -> int32_t {
        int32_t col = 0;
        while (col < ncolors && stamp[col] == v) col++;
        if (col == ncolors) { ncolors++; stamp.push_back(-1); load.push_back(0); }
        load[col]++;
        return col;
    };
    for (int64_t v = 0; v < nvar; v++) {
        if (!sampled[v]) continue;
        for_each_read(v, [&](int64_t b) {
            if (b != v && c.color[b] >= 0) stamp[c.color[b]] = v;
        });
        c.color[v] = pick(v);
    }
    lap("greedy colouring");
    // the greedy pass assumes reads are symmetric (true for compute_var_map output); verify, and
    // repair with explicit reverse-read lists when a raw index is asymmetric
    bool conflict = false;
    {
        std::vector<uint8_t> bad((size_t)compile_threads(), 0);
        parallel_for(nvar, [&](int64_t b0, int64_t b1, int t) {
            for (int64_t v = b0; v < b1 && !bad[(size_t)t]; v++) {
                if (!sampled[v]) continue;
                for_each_read(v, [&](int64_t b) {
                    if (b != v && c.color[b] == c.color[v]) bad[(size_t)t] = 1;
                });
            }
        });
        for (uint8_t x : bad) conflict = conflict || x;
    }
    if (conflict) {
        std::vector<int64_t> rcount(nvar + 1, 0);
        for (int64_t v = 0; v < nvar; v++)
            if (sampled[v]) for_each_read(v, [&](int64_t b) { if (b != v) rcount[b + 1]++; });
        for (int64_t v = 0; v < nvar; v++) rcount[v + 1] += rcount[v];
        std::vector<int32_t> readers((size_t)rcount[nvar]);
        std::vector<int64_t> fill(rcount.begin(), rcount.end() - 1);
        for (int64_t v = 0; v < nvar; v++)
            if (sampled[v]) for_each_read(v, [&](int64_t b) { if (b != v) readers[fill[b]++] = (int32_t)v; });
        std::fill(c.color.begin(), c.color.end(), -1);
        stamp.assign(1, -1);
        load.clear();
        ncolors = 0;
        for (int64_t v = 0; v < nvar; v++) {
            if (!sampled[v]) continue;
            for_each_read(v, [&](int64_t b) {
                if (b != v && c.color[b] >= 0) stamp[c.color[b]] = v;
            });
            for (int64_t j = rcount[v]; j < rcount[v + 1]; j++) {
                int32_t a = readers[j];
                if (c.color[a] >= 0) stamp[c.color[a]] = v;
            }
            c.color[v] = pick(v);
        }
    }

    lap("symmetry check");
    // fewer classes: iterated greedy (Culberson) -- recolour first fit with the vertices taken class
    // by class in a permuted class order; a class stays independent, so the count never grows, and
    // a few passes typically drop one or two classes (LR graph: 9 -> 7).  Every class costs a
    // kernel's latency floor, so this is sweep time.
    if (!conflict && ncolors > 2 && !diag_env("NSK_NO_RECOLOUR")) {
        std::vector<int32_t> newc(nvar), seq;
        seq.reserve((size_t)nvar);
        const int npass = diag_env("NSK_RECOLOUR_PASSES") ? atoi(diag_env("NSK_RECOLOUR_PASSES")) : 6;
        int stale = 0;                                   // passes in a row that dropped no class
        for (int pass = 0; pass < npass && stale < 2; pass++) {       // (each pass is a serial walk of the graph)
            std::vector<int64_t> size((size_t)ncolors, 0);
            for (int64_t v = 0; v < nvar; v++) if (c.color[v] >= 0) size[c.color[v]]++;
            std::vector<int32_t> cls((size_t)ncolors);
            for (int32_t k = 0; k < ncolors; k++) cls[k] = k;
            if (pass % 3 == 0) std::reverse(cls.begin(), cls.end());
            else std::stable_sort(cls.begin(), cls.end(), [&](int32_t a, int32_t b) {
                return pass % 3 == 1 ? size[a] > size[b] : size[a] < size[b]; });
            std::vector<int64_t> at((size_t)ncolors + 1, 0);           // counting sort by class rank
            std::vector<int32_t> rank((size_t)ncolors);
            for (int32_t r = 0; r < ncolors; r++) rank[cls[r]] = r;
            for (int32_t k = 0; k < ncolors; k++) at[rank[k] + 1] = size[k];
            for (int32_t r = 0; r < ncolors; r++) at[r + 1] += at[r];
            seq.assign((size_t)at[ncolors], 0);
            for (int64_t v = 0; v < nvar; v++) if (c.color[v] >= 0) seq[at[rank[c.color[v]]]++] = (int32_t)v;
            std::fill(newc.begin(), newc.end(), -1);
            // The vertices of one old class are not adjacent, so first fit gives each of them the same
            // colour whether they are taken one after the other or all at once: class by class, the
            // class's vertices over the host threads (each reads only colours of earlier classes).
            int32_t nnew = 0;
            for (int32_t r = 0; r < ncolors; r++) {
                const int64_t a0 = r ? at[r - 1] : 0, a1 = at[r];       // (at[] now holds the classes' ends in seq)
                std::vector<int32_t> tmax((size_t)compile_threads(), -1);
                parallel_for(a1 - a0, [&](int64_t b0, int64_t b1, int t) {
                    std::vector<int64_t> st((size_t)ncolors + 1, -1);
                    int32_t mx = -1;
                    for (int64_t i = a0 + b0; i < a0 + b1; i++) {
                        const int32_t v = seq[(size_t)i];
                        for_each_read(v, [&](int64_t b) {
                            if (b != v && newc[b] >= 0) st[newc[b]] = v;
                        });
                        int32_t col = 0;
                        while (st[col] == v) col++;                  // (at most ncolors colours are in use)
                        newc[v] = col;
                        mx = std::max(mx, col);
                    }
                    tmax[(size_t)t] = mx;
                });
                for (int32_t m : tmax) nnew = std::max(nnew, m + 1);
            }
            for (int64_t v = 0; v < nvar; v++) if (c.color[v] >= 0) c.color[v] = newc[v];
            stale = nnew < ncolors ? 0 : stale + 1;
            ncolors = nnew;
        }
        stamp.assign((size_t)ncolors, -1);
        load.assign((size_t)ncolors, 0);
        for (int64_t v = 0; v < nvar; v++) if (c.color[v] >= 0) load[c.color[v]]++;
    }

    lap("iterated greedy");
    // balancing: first fit leaves a few huge classes and a tail of tiny ones, and every class costs
    // a kernel's latency floor however few variables it holds.  Move variables, in id order, from
    // their class to the least populated class none of their neighbours is in (reads are symmetric
    // here -- the asymmetric repair above skips this pass).
    if (!conflict && ncolors > 2 && !diag_env("NSK_NO_BALANCE")) {
        for (int pass = 0; pass < 2; pass++)
            for (int64_t v = 0; v < nvar; v++) {
                if (!sampled[v]) continue;
                const int32_t cur = c.color[v];
                // (no class is more than one variable lighter than this one's: nothing below can move it, and its
                // neighbours need not be looked at -- most variables once the classes are level)
                int64_t lightest = load[0];
                for (int32_t k = 1; k < ncolors; k++) lightest = std::min(lightest, load[k]);
                if (lightest + 1 >= load[cur]) continue;
                for_each_read(v, [&](int64_t b) {
                    if (b != v && c.color[b] >= 0) stamp[c.color[b]] = v;
                });
                int32_t best = cur;
                for (int32_t k = 0; k < ncolors; k++)
                    if (k != cur && stamp[k] != v && load[k] + 1 < load[best]) best = k;
                if (best != cur) { load[cur]--; load[best]++; c.color[v] = best; }
            }
    }
    return ncolors;
}

// Entry-parallel groups (nsk_compile.h ep_desc): the general tiles of an EP colour, four at a time, as rows of 64
// list entries sorted by their member count.  general_words(v, &out) is the compiler's per-variable entry list;
// lap(name) closes a timed stage.
template <typename WordsFn, typename LapFn>
static int build_ep_groups(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, WordsFn &&general_words, LapFn &&lap,
                           bool verbose, std::string &err) {
    const int64_t nw = c.nweight;
    (void)d; (void)nw;
    // ---- entry-parallel groups (nsk_compile.h ep_desc): the general tiles of an EP colour, four at a
    // time, as rows of 64 list entries sorted by their member count
    c.phase_ep_base.assign((size_t)ncolors + 1, 0);
    for (int32_t k = 0; k < ncolors; k++) {
        const int64_t ngt = (c.phase_wb_base[k + 1] - c.phase_wb_base[k]) - c.phase_gen_tile[k];
        c.phase_ep_base[k + 1] = c.phase_ep_base[k] + (c.phase_ep[k] ? (ngt + 3) / 4 : 0);
    }
    {
        const int64_t ngroups = c.phase_ep_base[ncolors];
        c.ep_desc.assign((size_t)ngroups * 4 + 4, 0u);
        c.ep_wrow.assign((size_t)ngroups + 1, 0u);
        std::vector<int32_t> group_colour((size_t)ngroups);
        for (int32_t k = 0; k < ncolors; k++)
            for (int64_t gi = c.phase_ep_base[k]; gi < c.phase_ep_base[k + 1]; gi++) group_colour[gi] = k;
        auto group_range = [&](int64_t gi, int64_t &p0, int64_t &p1) {
            const int32_t k = group_colour[gi];
            p0 = c.phase_start[k] + 64 * (c.phase_gen_tile[k] + 4 * (gi - c.phase_ep_base[k]));
            p1 = std::min(p0 + 256, c.phase_fast_end[k]);
        };
        std::vector<uint64_t> subrows((size_t)ngroups + 1, 0);
        // row classes: member count M = 0..3 of the entries with ordinal < 8 ("base", classes 0-3),
        // then the same for ordinals 8..15 ("overflow", classes 4-7): the kernels hold 8 list positions
        // per variable in LDS and take a group with longer lists in two passes
        auto row_class = [](uint32_t m, uint32_t ordinal) { return m + (ordinal >= 8 ? 4u : 0u); };
        parallel_for(ngroups, [&](int64_t g0, int64_t g1, int) {          // pass A: rows per class
            std::vector<uint32_t> w;
            for (int64_t gi = g0; gi < g1; gi++) {
                int64_t p0, p1;
                group_range(gi, p0, p1);
                uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, emax = 0, maxcard = 2;
                for (int64_t p = p0; p < p1; p++) {
                    if (c.p_vid[p] < 0) continue;
                    general_words(c.p_vid[p], &w);
                    uint32_t ne = 0;
                    for (size_t j = 0; j < w.size(); j += 2 + ((w[j + 1] >> 4) & 7u)) { cnt[row_class((w[j + 1] >> 4) & 7u, ne)]++; ne++; }
                    emax = std::max(emax, ne);
                    maxcard = std::max(maxcard, (uint32_t)d->variable[c.p_vid[p]].cardinality);
                }
                uint32_t *gd = &c.ep_desc[(size_t)gi * 4];
                uint64_t sr = 0;
                gd[1] = 0; gd[3] = 0;
                for (uint32_t cl = 0; cl < 8; cl++) {
                    const uint32_t rows = (cnt[cl] + 63) / 64;
                    gd[cl < 4 ? 1 : 3] |= rows << (8 * (cl & 3u));
                    sr += (uint64_t)rows * (2 + (cl & 3u));
                }
                gd[2] = emax | (maxcard << 8);
                subrows[gi + 1] = sr;
                uint32_t nrows = 0;
                for (uint32_t cl = 0; cl < 8; cl++) nrows += (cnt[cl] + 63) / 64;
                c.ep_wrow[gi + 1] = nrows;
            }
        }, 8);                                      // (a group is 256 variables' worth of work)
        lap("entry-parallel groups: rows");
        for (int64_t gi = 0; gi < ngroups; gi++) {
            const uint64_t next = (uint64_t)c.ep_wrow[gi] + c.ep_wrow[gi + 1];
            if (next >= ((uint64_t)1 << 31)) { err = "entry-parallel stream too large"; return NSK_E_RANGE; }
            c.ep_wrow[gi + 1] = (uint32_t)next;
        }
        for (int64_t gi = 0; gi < ngroups; gi++) subrows[gi + 1] += subrows[gi];
        if (subrows[ngroups] * 64 >= ((uint64_t)1 << 31)) { err = "entry-parallel stream too large"; return NSK_E_RANGE; }
        c.ep_adj.assign((size_t)subrows[ngroups] * 64 + 64, 0u);
        // structural visit counts (nsk_compile.h ep_kstat): global accumulators only (graphs with few
        // weights accumulate in LDS tables, where an update costs nothing); counted in pass B (atomic
        // increments: a weight's entries are spread over the groups, contention is negligible)
        const bool want_kstat = ngroups > 0 && nw > 256 && (int64_t)ncolors * nw * 2 <= ((int64_t)1 << 26) && !diag_env("NSK_NO_KSTAT") &&
                                c.ndirect == 0;      // (direct weights are updated at their visit: every visit must reach the kernel)
        if (want_kstat) c.ep_kstat.assign((size_t)ncolors * 2 * (size_t)nw, 0u);
        lap("entry-parallel groups: allocation");
        parallel_for(ngroups, [&](int64_t g0, int64_t g1, int) {          // pass B: fill
            std::vector<uint32_t> w;
            for (int64_t gi = g0; gi < g1; gi++) {
                int64_t p0, p1;
                group_range(gi, p0, p1);
                uint32_t *gd = &c.ep_desc[(size_t)gi * 4];
                gd[0] = (uint32_t)subrows[gi];
                uint64_t base[8], at[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // first sub-row / entries placed, per class
                uint64_t sr = subrows[gi];
                for (uint32_t cl = 0; cl < 8; cl++) {
                    const uint32_t m = cl & 3u;
                    base[cl] = sr;
                    const uint32_t rows = (gd[cl < 4 ? 1 : 3] >> (8 * m)) & 255u;
                    // padding entries of the last row: owned by no candidate, empty member slots
                    for (uint64_t r = 0; r < rows; r++)
                        for (uint32_t e = 0; e < 64; e++) {
                            uint32_t *row = &c.ep_adj[(sr + r * (2 + m)) * 64];
                            row[2 * e] = 0u; row[2 * e + 1] = 14u << 14;
                            for (uint32_t mm = 0; mm < m; mm++) row[(2 + mm) * 64 + e] = NSK_GEN_NULL;
                        }
                    sr += (uint64_t)rows * (2 + m);
                }
                const int32_t gk = group_colour[gi];
                for (int64_t p = p0; p < p1; p++) {
                    if (c.p_vid[p] < 0) continue;
                    general_words(c.p_vid[p], &w);
                    const nsk_variable &var = d->variable[c.p_vid[p]];
                    if (want_kstat && var.dataType == 0) {
                        const size_t o = var.isEvidence == 1 ? 0 : 1;
                        for (size_t j = 0; j < w.size(); j += 2 + ((w[j + 1] >> 4) & 7u))
                            if (!c.w_fixed[w[j]])
                                __atomic_fetch_add(&c.ep_kstat[((size_t)gk * 2 + o) * (size_t)nw + w[j]], 1u, __ATOMIC_RELAXED);
                    }
                    uint32_t ordinal = 0;
                    for (size_t j = 0; j < w.size(); ordinal++) {
                        const uint32_t m = (w[j + 1] >> 4) & 7u, cl = row_class(m, ordinal);
                        const uint64_t r = at[cl] / 64, e = at[cl] % 64;
                        at[cl]++;
                        uint32_t *row = &c.ep_adj[(base[cl] + r * (2 + m)) * 64];
                        const uint32_t wid = w[j];
                        row[2 * e] = wid | (ordinal << 27);
                        row[2 * e + 1] = w[j + 1] | ((uint32_t)(p - p0) << 23) | (c.w_fixed[wid] ? 0x80000000u : 0u);
                        for (uint32_t mm = 0; mm < m; mm++)
                            row[(2 + mm) * 64 + e] = (uint32_t)c.iid[w[j + 2 + mm] & NSK_GEN_NULL] | (w[j + 2 + mm] & ~NSK_GEN_NULL);
                        j += 2 + m;
                    }
                }
            }
        }, 8);
        // ---- value windows (nsk_compile.h ep_win): per group the 16-byte chunks of the value array its members
        // lie in; member ids inside the kept chunks become offsets into the group's LDS copy
        c.ep_win.clear();
        c.ep_win_off.assign((size_t)ngroups + 1, 0u);
#ifdef NSK_EP_WIN
        const bool want_win = ngroups > 0 && c.vbytes == 1 && !diag_env("NSK_NO_EP_WIN") && c.nid < (int64_t)NSK_EP_WIN_BASE;
#else
        const bool want_win = false;            // (measured and not kept: nsk_compile.h ep_win)
#endif
        if (want_win) {
            std::vector<std::vector<uint32_t>> kept((size_t)ngroups);
            std::vector<int64_t> st((size_t)compile_threads() * 2, 0);     // members in a window / members
            parallel_for(ngroups, [&](int64_t g0, int64_t g1, int t) {
                std::vector<uint32_t> ch;
                std::vector<std::pair<uint32_t, uint32_t>> cnt;        // (uses, chunk)
                for (int64_t gi = g0; gi < g1; gi++) {
                    const uint32_t *gd = &c.ep_desc[(size_t)gi * 4];
                    // every member word of the group's rows
                    ch.clear();
                    auto each_member = [&](auto &&fn) {
                        uint64_t sr = subrows[gi];
                        for (uint32_t cl = 0; cl < 8; cl++) {
                            const uint32_t m = cl & 3u, rows = (gd[cl < 4 ? 1 : 3] >> (8 * m)) & 255u;
                            for (uint64_t r = 0; r < rows; r++)
                                for (uint32_t mm = 0; mm < m; mm++) {
                                    uint32_t *row = &c.ep_adj[(sr + r * (2 + m) + 2 + mm) * 64];
                                    for (uint32_t e = 0; e < 64; e++) if ((row[e] & NSK_GEN_NULL) != NSK_GEN_NULL) fn(row[e]);
                                }
                            sr += (uint64_t)rows * (2 + m);
                        }
                    };
                    each_member([&](uint32_t &wd) { ch.push_back((wd & NSK_GEN_NULL) >> 4); });
                    std::sort(ch.begin(), ch.end());
                    cnt.clear();
                    for (size_t i = 0; i < ch.size();) {
                        size_t j = i;
                        while (j < ch.size() && ch[j] == ch[i]) j++;
                        cnt.emplace_back((uint32_t)(j - i), ch[i]);
                        i = j;
                    }
                    if (cnt.size() > NSK_EP_WIN_CHUNKS) {                // keep the most used chunks (ties: lowest id)
                        std::sort(cnt.begin(), cnt.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) {
                            return a.first != b.first ? a.first > b.first : a.second < b.second; });
                        cnt.resize(NSK_EP_WIN_CHUNKS);
                    }
                    std::vector<uint32_t> &kp = kept[(size_t)gi];
                    kp.clear();
                    for (const auto &x : cnt) kp.push_back(x.second);
                    std::sort(kp.begin(), kp.end());
                    each_member([&](uint32_t &wd) {
                        const uint32_t id = wd & NSK_GEN_NULL;
                        const auto it = std::lower_bound(kp.begin(), kp.end(), id >> 4);
                        st[2 * (size_t)t + 1]++;
                        if (it == kp.end() || *it != (id >> 4)) return;
                        wd = (wd & ~NSK_GEN_NULL) | (NSK_EP_WIN_BASE + (uint32_t)(it - kp.begin()) * 16u + (id & 15u));
                        st[2 * (size_t)t]++;
                    });
                }
            }, 8);
            for (int64_t gi = 0; gi < ngroups; gi++) c.ep_win_off[gi + 1] = c.ep_win_off[gi] + (uint32_t)kept[(size_t)gi].size();
            c.ep_win.resize((size_t)c.ep_win_off[ngroups]);
            parallel_for(ngroups, [&](int64_t g0, int64_t g1, int) {
                for (int64_t gi = g0; gi < g1; gi++)
                    std::copy(kept[(size_t)gi].begin(), kept[(size_t)gi].end(), c.ep_win.begin() + c.ep_win_off[gi]);
            }, 64);
            if (verbose) {
                int64_t in = 0, all = 0;
                for (size_t t = 0; t < st.size(); t += 2) { in += st[t]; all += st[t + 1]; }
                fprintf(stderr, "[nsk] value windows: %.1f chunks per group, %.2f %% of %lld members inside\n",
                        (double)c.ep_win.size() / (double)ngroups, all ? 100.0 * (double)in / (double)all : 0.0, (long long)all);
            }
        }
        lap("entry-parallel groups: value windows");
        if (verbose && ngroups)
            fprintf(stderr, "[nsk] entry-parallel groups %lld, stream %.1f MB\n", (long long)ngroups,
                    (double)subrows[ngroups] * 256 / 1e6);
    }
    return NSK_OK;
}

// Implicit adjacency of table segments (nsk_compile.h seg_aff): per tile the base ids of its member runs when the
// lanes' members are consecutive (a regular grid), so the sweep kernels need no stream there.
static int build_segment_adjacency(Compiled &c, std::string &err) {
    uint64_t ntile4 = 0;
    const bool no_aff = diag_env("NSK_NO_AFFINE") != nullptr;
    for (Compiled::Segment &sg : c.segments) {
        sg.aff = -1;
        if (sg.ztab < 0 || no_aff) continue;
        sg.aff = (int64_t)ntile4;
        ntile4 += (uint64_t)sg.ntiles * (sg.nslots > 4 ? 2 : 1);
    }
    if (ntile4 >= ((uint64_t)1 << 30)) { err = "implicit adjacency table too large"; return NSK_E_RANGE; }
    c.seg_aff.assign((size_t)ntile4 * 4 + 4, 0xFFFFFFFFu);
    for (const Compiled::Segment &sg : c.segments) {
        if (sg.aff < 0) continue;
        const int nch = sg.nslots > 4 ? 2 : 1;
        parallel_for(sg.ntiles, [&](int64_t tb0, int64_t tb1, int) {
            for (int64_t t = tb0; t < tb1; t++) {
                const uint64_t wbase = ((uint64_t)sg.adj_off + (uint64_t)t * 64 * nch) * 4;
                int64_t first = -1;                         // first live lane
                for (int64_t i = 0; i < 64 && first < 0; i++) if (c.p_vid[sg.pos0 + 64 * t + i] >= 0) first = i;
                if (first < 0) continue;
                bool ok = true;
                uint32_t base[8];
                for (uint32_t j = 0; j < (uint32_t)(4 * nch) && ok; j++) {
                    const uint64_t wj = wbase + 256 * (j / 4) + (j % 4);
                    const int64_t b0 = (int64_t)c.adj[wj + 4 * first] - first;
                    if (b0 < 0 || b0 + 63 >= c.nid) { ok = false; break; }       // every lane reads a valid id
                    for (int64_t i = 0; i < 64 && ok; i++)
                        if (c.p_vid[sg.pos0 + 64 * t + i] >= 0 && (int64_t)c.adj[wj + 4 * i] != b0 + i) ok = false;
                    base[j] = (uint32_t)b0;
                }
                if (!ok || base[0] == 0xFFFFFFFFu) continue;
                for (int cidx = 0; cidx < nch; cidx++)
                    for (int q = 0; q < 4; q++) c.seg_aff[((size_t)sg.aff + (size_t)t * nch + cidx) * 4 + q] = base[4 * cidx + q];
            }
        }, 64);
    }
    return NSK_OK;
}

// Wide quads pay from a few hundred thousand variables per handle on (one MI355X, us per sweep, tile-by-tile kernel /
// wide-quad kernel, tools/sessions/r6_s25.sh: 256 x 256 grid 6.63 / 6.60, 512 x 512 6.82 / 7.10, 500 x 1000 7.30 / 6.33, 1M grid
// 9.00 / 6.64, 4M 13.8 / 10.2-11.9, 10M 23.9 / 15.4) -- since the launch's few quads that are NOT wide are sampled by workgroups
// of their own (TabwCold.rest); while the waves whose turn they were sampled them in line, the bound was 3M (1M grid 9.1
// against 10.9).  NSK_DIAG=1 NSK_WIDE_MIN=n moves the bound (the small-grid tests use 0).
static int64_t wide_min_variables() {
    const char *e = diag_env("NSK_WIDE_MIN");
    return e ? atoll(e) : 400000;
}

// Wide quads of table segments (nsk_compile.h seg_wide): per quad the slot bases when one lane can take four
// consecutive positions, plus the few positions whose member lies elsewhere (exceptions).
static int build_segment_wide(Compiled &c, std::string &err) {
    c.seg_wide.clear();
    c.wide_exc.clear();
    c.ntab_quads = c.nwide_quads = 0;
    for (Compiled::Segment &sg : c.segments) sg.wide = -1;
    if (c.vbytes != 1 || diag_env("NSK_NO_WIDE") || c.nsampled < wide_min_variables()) {
        c.seg_wide.assign(4, 0xFFFFFFFFu); c.wide_exc.assign(2, 0u); return NSK_OK;
    }
    uint64_t ndw = 0;
    for (Compiled::Segment &sg : c.segments) {
        if (sg.ztab < 0) continue;
        const int nch = sg.nslots > 4 ? 2 : 1;
        const int64_t nq = ((sg.pos0 + 64 * (int64_t)sg.ntiles + 255) >> 8) - (sg.pos0 >> 8);
        sg.wide = (int64_t)ndw;
        ndw += (uint64_t)nq * NSK_WIDE_STRIDE(nch);
        c.ntab_quads += nq;
    }
    if (ndw >= ((uint64_t)1 << 31)) { err = "wide-quad table too large"; return NSK_E_RANGE; }
    c.seg_wide.assign((size_t)ndw + 4, 0xFFFFFFFFu);
    const int T = compile_threads();
    std::vector<std::vector<uint32_t>> exc_of((size_t)T);              // per thread: {descriptor dword, count, pairs ...}
    std::vector<int64_t> nwide_of((size_t)T, 0);
    for (const Compiled::Segment &sg : c.segments) {
        if (sg.wide < 0) continue;
        const int nch = sg.nslots > 4 ? 2 : 1, stride = NSK_WIDE_STRIDE(nch);
        const int64_t q0 = sg.pos0 >> 8;
        const int64_t nq = ((sg.pos0 + 64 * (int64_t)sg.ntiles + 255) >> 8) - q0;
        parallel_for(nq, [&](int64_t qb0, int64_t qb1, int th) {
            std::vector<uint32_t> &exo = exc_of[(size_t)th];
            for (int64_t qi = qb0; qi < qb1; qi++) {
                const int64_t P = (q0 + qi) << 8;                       // the quad's first position
                if (P < sg.pos0 || P + 256 > sg.pos0 + 64 * (int64_t)sg.ntiles) continue;     // not wholly inside the segment
                const int64_t t0 = (P - sg.pos0) >> 6;
                // member id of slot j at offset o of the quad
                auto member = [&](int64_t o, uint32_t j) -> int64_t {
                    const uint64_t wbase = ((uint64_t)sg.adj_off + (uint64_t)(t0 + (o >> 6)) * 64 * nch) * 4;
                    return (int64_t)c.adj[wbase + 256 * (j / 4) + (j % 4) + 4 * (uint64_t)(o & 63)];
                };
                int64_t first = -1, last = -1;
                for (int64_t o = 0; o < 256; o++)
                    if (c.p_vid[P + o] >= 0) { if (first < 0) first = o; last = o; }
                if (first < 0) continue;
                uint32_t base[8], smask = 0, nexc = 0, exc[2 * NSK_WIDE_MAXEXC];
                bool ok = true;
                for (uint32_t j = 0; j < sg.nslots && ok; j++) {
                    // a slot that names the always-zero id in every lane is no member at all
                    bool zero = true;
                    for (int64_t o = first; o <= last && zero; o++)
                        if (c.p_vid[P + o] >= 0 && member(o, j) != c.zero_id) zero = false;
                    if (zero) { base[j] = 0xFFFFFFFFu; continue; }
                    // the base most live positions agree on: the first's or the last's (an odd cell sits at a run's end)
                    int64_t best = -1, best_miss = 1 << 30;
                    const int64_t cand[3] = {member(first, j) - first, member(last, j) - last,
                                             member((first + last) / 2, j) - (first + last) / 2};
                    for (int k = 0; k < 3; k++) {
                        const int64_t b = cand[k];
                        if (b < 0 || b + 255 >= c.nid || (k > 0 && b == cand[0]) || (k > 1 && b == cand[1])) continue;
                        int64_t miss = 0;
                        for (int64_t o = first; o <= last && miss <= NSK_WIDE_MAXEXC; o++)
                            if (c.p_vid[P + o] >= 0 && member(o, j) != b + o) miss++;
                        if (miss < best_miss) { best_miss = miss; best = b; }
                    }
                    if (best < 0 || nexc + best_miss > NSK_WIDE_MAXEXC) {
                        if (getenv("NSK_DEBUG_WIDE"))
                            fprintf(stderr, "[nsk] quad at %lld (segment pos0 %lld): slot %u best %lld misses %lld (first %lld last %lld cand %lld %lld %lld)\n",
                                    (long long)P, (long long)sg.pos0, j, (long long)best, (long long)best_miss, (long long)first, (long long)last,
                                    (long long)cand[0], (long long)cand[1], (long long)cand[2]);
                        if (getenv("NSK_DEBUG_WIDE")) {
                            for (int64_t o = first; o <= last; o++)
                                if (c.p_vid[P + o] >= 0 && member(o, j) != best + o) fprintf(stderr, " [o %lld vid %d member %lld]", (long long)o, c.p_vid[P + o], (long long)member(o, j));
                            fprintf(stderr, "\n");
                        }
                        ok = false; break; }
                    base[j] = (uint32_t)best;
                    smask |= 1u << j;
                    for (int64_t o = first; o <= last; o++)
                        if (c.p_vid[P + o] >= 0 && member(o, j) != best + o) {
                            exc[2 * nexc] = (uint32_t)o | (j << 8);
                            exc[2 * nexc + 1] = (uint32_t)member(o, j);
                            nexc++;
                        }
                }
                if (!ok || smask == 0) continue;
                uint32_t any = 0;
                for (uint32_t j = 0; j < sg.nslots; j++) if ((smask >> j) & 1u) { any = base[j]; break; }
                uint32_t *dq = &c.seg_wide[(size_t)sg.wide + (size_t)qi * stride];
                for (uint32_t j = 0; j < (uint32_t)(4 * nch); j++) dq[j] = (j < sg.nslots && ((smask >> j) & 1u)) ? base[j] : any;
                if (dq[0] == 0xFFFFFFFFu) { for (uint32_t j = 0; j < (uint32_t)(4 * nch); j++) dq[j] = 0xFFFFFFFFu; continue; }   // (cannot happen: ids < 2^31)
                dq[4 * nch] = 0; dq[4 * nch + 1] = nexc; dq[4 * nch + 2] = smask; dq[4 * nch + 3] = 0;
                nwide_of[(size_t)th]++;
                if (nexc) {
                    exo.push_back((uint32_t)(sg.wide + qi * stride));
                    exo.push_back(nexc);
                    exo.insert(exo.end(), exc, exc + 2 * nexc);
                }
            }
        }, 16);
        // the segment's exception lists: the threads hold ascending ranges of its quads, so thread order is quad
        // order whatever the thread count
        for (int th = 0; th < T; th++) {
            std::vector<uint32_t> &exo = exc_of[(size_t)th];
            for (size_t i = 0; i < exo.size();) {
                const uint32_t dq = exo[i], n = exo[i + 1];
                c.seg_wide[(size_t)dq + 4 * nch] = (uint32_t)(c.wide_exc.size() / 2);
                c.wide_exc.insert(c.wide_exc.end(), exo.begin() + (long)i + 2, exo.begin() + (long)i + 2 + 2 * (long)n);
                i += 2 + 2 * (size_t)n;
            }
            exo.clear();
        }
    }
    c.nwide_quads = 0;
    for (int th = 0; th < T; th++) c.nwide_quads += nwide_of[(size_t)th];
    c.wide_exc.resize(c.wide_exc.size() + 2, 0u);
    return NSK_OK;
}

// Hub streams: the long-list variables a whole wave (or workgroup) samples, laid out like the entry-parallel rows.
// general_words(v, &out, hub, cap) is the compiler's per-variable entry list.
template <typename WordsFn>
static int build_hub_streams(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, WordsFn &&general_words, bool no_general,
                             bool verbose, std::string &err) {
    const int64_t nw = c.nweight, nvar = c.nvar;
    (void)d; (void)nw; (void)nvar; (void)err; (void)verbose;
    // ---- entry-parallel hub streams: a hub (a long-list variable sampled by a whole wave) whose
    // factors are all of the general-tile kind gets its entries laid out one per LANE -- word j of
    // entry e of round r at hub_adj[off + (r * (2 + M) + j) * 64 + e] -- so that one coalesced row
    // load per word, one gather per member and a list-order sum over the lanes replace the
    // dependent fidx -> factor -> edge -> value chain of the generic hub walk.
    c.phase_hub_base.assign((size_t)ncolors + 1, 0);            // descriptors: hub ranges only, colour-major
    for (int32_t k = 0; k < ncolors; k++)
        c.phase_hub_base[k + 1] = c.phase_hub_base[k] + (c.phase_heavy_end[k] - c.phase_fast_end[k]);
    c.hub_desc.assign((size_t)(c.phase_hub_base[ncolors] + 1) * 4, 0u);
    c.phase_bighub_base.assign((size_t)ncolors + 1, 0);
    std::vector<int32_t> hub_colour;
    if (!diag_env("NSK_NO_HUB_EP") && !no_general) {
        std::vector<int64_t> hubs;
        for (int32_t k = 0; k < ncolors; k++)
            for (int64_t p = c.phase_fast_end[k]; p < c.phase_heavy_end[k]; p++)
                if (c.p_vid[p] >= 0) { hubs.push_back(p); hub_colour.push_back(k); }
        std::vector<uint32_t> nent(hubs.size(), 0), mh(hubs.size(), 0);
        parallel_for((int64_t)hubs.size(), [&](int64_t b0, int64_t b1, int) {
            std::vector<uint32_t> w;
            for (int64_t h = b0; h < b1; h++) {
                // (a colour laid out as entry-parallel groups has the block-per-hub kernels for long lists)
                if (!general_words(c.p_vid[hubs[h]], &w, true, c.phase_ep[hub_colour[h]] ? 16384 : 256)) continue;
                uint32_t ne = 0, mo = 0;
                for (size_t j = 0; j < w.size(); j += 2 + ((w[j + 1] >> 4) & 7u)) { ne++; mo = std::max(mo, (w[j + 1] >> 4) & 7u); }
                nent[h] = ne; mh[h] = mo;
            }
        }, 4);
        uint64_t total = 0;
        std::vector<uint64_t> off(hubs.size(), 0);
        for (size_t h = 0; h < hubs.size(); h++) {
            if (!nent[h]) continue;
            off[h] = total;
            total += (uint64_t)((nent[h] + 63) / 64) * (2 + mh[h]) * 64;
        }
        if (total < ((uint64_t)1 << 31)) {
            c.hub_adj.assign((size_t)total + 64, 0u);
            parallel_for((int64_t)hubs.size(), [&](int64_t b0, int64_t b1, int) {
                std::vector<uint32_t> w;
                for (int64_t h = b0; h < b1; h++) {
                    if (!nent[h]) continue;
                    const int64_t p = hubs[h];
                    const nsk_variable &var = d->variable[c.p_vid[p]];
                    general_words(c.p_vid[p], &w, true, 16384);
                    const uint32_t rows = 2 + mh[h], rounds = (nent[h] + 63) / 64;
                    uint32_t *base = &c.hub_adj[off[h]];
                    for (uint32_t r = 0; r < rounds; r++)              // padding entries: owned by no candidate
                        for (uint32_t e = 0; e < 64; e++) {
                            base[(r * rows + 0) * 64 + e] = 0u;
                            base[(r * rows + 1) * 64 + e] = 14u << 14;
                            for (uint32_t m = 0; m < mh[h]; m++) base[(r * rows + 2 + m) * 64 + e] = NSK_GEN_NULL;
                        }
                    uint32_t e = 0;
                    for (size_t j = 0; j < w.size(); e++) {
                        const uint32_t no = (w[j + 1] >> 4) & 7u, r = e / 64, l = e % 64;
                        base[(r * rows + 0) * 64 + l] = w[j];
                        base[(r * rows + 1) * 64 + l] = w[j + 1];
                        for (uint32_t m = 0; m < no; m++)
                            base[(r * rows + 2 + m) * 64 + l] = (uint32_t)c.iid[w[j + 2 + m] & NSK_GEN_NULL] | (w[j + 2 + m] & ~NSK_GEN_NULL);
                        j += 2 + no;
                    }
                    const int32_t hk = hub_colour[h];
                    uint32_t *hd = &c.hub_desc[(size_t)(c.phase_hub_base[hk] + (p - c.phase_fast_end[hk])) * 4];
                    // hd[3] = 1: a long list, evaluated by a whole workgroup (k_gibbs_ep / k_learn_ep)
                    hd[0] = (uint32_t)off[h]; hd[1] = nent[h]; hd[2] = mh[h] | ((uint32_t)var.cardinality << 8);
                    hd[3] = (c.phase_ep[hk] && nent[h] > 128) ? 1u : 0u;
                }
            }, 4);
            for (size_t h = 0; h < hubs.size(); h++) {          // (hubs are listed colour by colour)
                const int32_t hk = hub_colour[h];
                if (!c.hub_desc[(size_t)(c.phase_hub_base[hk] + (hubs[h] - c.phase_fast_end[hk])) * 4 + 3]) continue;
                c.bighub_pos.push_back((uint32_t)hubs[h]);
                c.phase_bighub_base[hk + 1]++;
            }
            c.nhub_ep = 0;
            for (size_t h = 0; h < hubs.size(); h++) if (nent[h]) c.nhub_ep++;
            if (verbose) fprintf(stderr, "[nsk] hubs %zu, entry-parallel %lld, stream %.1f MB\n", hubs.size(),
                                 (long long)c.nhub_ep, (double)total * 4 / 1e6);
        }
    }
    for (int32_t k = 0; k < ncolors; k++) c.phase_bighub_base[k + 1] += c.phase_bighub_base[k];
    if (c.bighub_pos.empty()) c.bighub_pos.push_back(0);
    return NSK_OK;
}

// Learning launches over homogeneous segments: segments grouped by (kind, chunks) into tables of <= 8, the
// NSK_LEARN_SEG_LAUNCHES largest tables of a colour become launches, the tiles of the others join the colour's
// learning rest list.
static void plan_learning_launches(Compiled &c, int32_t ncolors) {
    // learning launches: segments grouped by (kind, chunks) into tables of <= 8, the
    // NSK_LEARN_SEG_LAUNCHES largest tables of a colour become launches, the tiles of the
    // others join the colour's rest list
    c.phase_learn_rest_base.assign((size_t)ncolors + 1, 0);
    for (int32_t k = 0; k < ncolors; k++) {
        std::vector<Compiled::SegLaunch> tabs;
        for (int tab = 0; tab <= 1; tab++)                  // 0 no draw table, 1 table (compact stream or not)
        for (int kind = 0; kind <= 4; kind++)
            for (int nch = 1; nch <= 2; nch++) {
                Compiled::SegLaunch t;
                memset(&t, 0, sizeof(t));
                t.phase = k; t.kind = tab ? 8 : kind; t.nch = nch; t.tab = tab;
                std::vector<const Compiled::Segment *> mine;       // largest first (seg_of_tile's first probe)
                for (const Compiled::Segment &sg : c.segments) {
                    // table segments of any function share a launch (the table encodes the function)
                    if (sg.phase != k || (sg.nslots > 4 ? 2 : 1) != nch || (sg.ztab < 0 ? 0 : 1) != tab ||
                        (tab ? kind != 0 : (int)(sg.kind == 1 ? 3 : sg.kind) != kind))
                        continue;
                    mine.push_back(&sg);
                }
                std::stable_sort(mine.begin(), mine.end(), [](const Compiled::Segment *a, const Compiled::Segment *b) {
                    return a->ntiles > b->ntiles; });
                for (const Compiled::Segment *sgp : mine) {
                    const Compiled::Segment &sg = *sgp;
                    t.pos0[t.n] = (int32_t)sg.pos0; t.adj_off[t.n] = sg.adj_off; t.prog[t.n] = sg.prog;
                    t.aff[t.n] = sg.aff >= 0 ? (uint32_t)sg.aff : 0xFFFFFFFFu;
                    t.zoff[t.n] = sg.ztab >= 0 ? (uint32_t)sg.ztab : 0u;
                    t.zmask[t.n] = (1u << sg.nslots) - 1u;
                    t.ev[t.n] = sg.ev;
                    t.wide[t.n] = sg.wide;
                    t.tile_start[t.n + 1] = t.tile_start[t.n] + sg.ntiles;
                    if (++t.n == 8) { tabs.push_back(t); t.n = 0; t.tile_start[0] = 0; }
                }
                if (t.n) tabs.push_back(t);
            }
        std::stable_sort(tabs.begin(), tabs.end(), [](const Compiled::SegLaunch &a, const Compiled::SegLaunch &b) {
            return a.tile_start[a.n] > b.tile_start[b.n]; });
        std::vector<uint32_t> extra;
        for (size_t i = 0; i < tabs.size(); i++) {
            if (i < NSK_LEARN_SEG_LAUNCHES && !diag_env("NSK_NO_LEARN_SEG")) { c.learn_seg.push_back(tabs[i]); continue; }
            for (int j = 0; j < tabs[i].n; j++)
                for (int32_t t = 0; t < tabs[i].tile_start[j + 1] - tabs[i].tile_start[j]; t++)
                    extra.push_back((uint32_t)((tabs[i].pos0[j] - c.phase_start[k]) / 64 + t));
        }
        for (int64_t i = c.phase_rest_base[k]; i < c.phase_rest_base[k + 1]; i++) extra.push_back(c.rest_tiles[i]);
        std::sort(extra.begin(), extra.end());
        c.learn_rest_tiles.insert(c.learn_rest_tiles.end(), extra.begin(), extra.end());
        c.phase_learn_rest_base[k + 1] = (int64_t)c.learn_rest_tiles.size();
    }
    if (c.learn_rest_tiles.empty()) c.learn_rest_tiles.push_back(0);
}

// Homogeneous segments: runs of uniform tiles with one program (and one evidence flag) become segment launches,
// with a draw table when their members are binary; the other uniform / shape tiles of a colour form its rest
// list.  Also the colour's all-binary general tiles and its tiles with per-lane headers.  lane_words(v, out) is
// the compiler's per-variable word list of the fast path.
template <typename LaneWordsFn>
static int plan_segments(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, LaneWordsFn &&lane_words,
                         const std::vector<uint8_t> &fast, bool verbose, std::string &err) {
    const int64_t nw = c.nweight, nvar = c.nvar;
    (void)d; (void)nw; (void)nvar; (void)err; (void)verbose;
    c.phase_gen_bin_tile.assign((size_t)ncolors, 0);
    for (int32_t k = 0; k < ncolors; k++) {
        int64_t t = c.phase_wb_base[k + 1] - c.phase_wb_base[k];
        while (t > c.phase_gen_tile[k] && ((c.tiles[4 * (c.phase_wb_base[k] + t - 1) + 3] >> 12) & 15u) <= 2u) t--;
        c.phase_gen_bin_tile[k] = t;
    }
    c.phase_dyn_base.assign((size_t)ncolors + 1, 0);
    for (int32_t k = 0; k < ncolors; k++) {
        for (int64_t b = 0; b < c.phase_wb_base[k + 1] - c.phase_wb_base[k]; b++)
            if (c.tiles[4 * (c.phase_wb_base[k] + b) + 2] == 0xFFFFFFFFu)
                c.dyn_tiles.push_back((uint32_t)(c.phase_start[k] + 64 * b));
        c.phase_dyn_base[k + 1] = (int64_t)c.dyn_tiles.size();
    }
    if (c.dyn_tiles.empty()) c.dyn_tiles.push_back(0);
    // homogeneous segments and the rest list
    const int64_t SEG_MIN = 1;
    std::map<uint32_t, int64_t> ztab_of;                        // program -> first table entry
    c.phase_rest_base.assign((size_t)ncolors + 1, 0);
    for (int32_t k = 0; k < ncolors; k++) {
        const int64_t nt = c.phase_wb_base[k + 1] - c.phase_wb_base[k];
        auto tile_ev = [&](int64_t b, bool &full) -> int {     // common isEvidence of a tile or -999
            const int64_t p0 = c.phase_start[k] + 64 * b, p1 = std::min(p0 + 64, c.phase_fast_end[k]);
            full = true;                                       // padding lanes are masked in-kernel
            int ev = -999;
            bool any = false;
            for (int64_t p = p0; p < p1; p++) {
                if (c.p_vid[p] < 0) continue;
                const int e2 = d->variable[c.p_vid[p]].isEvidence;
                any = true;
                if (ev == -999) ev = e2;
                else if (e2 != ev) return -999;
            }
            // (a tile of padding positions only -- run padding, place_variables -- goes with the tiles in front of it)
            if (!any)
                for (int64_t q = p0 - 1; q >= c.phase_start[k]; q--)
                    if (c.p_vid[q] >= 0) return (int)d->variable[c.p_vid[q]].isEvidence;
            return ev;
        };
        int64_t b = 0;
        while (b < nt) {
            const uint32_t *td = &c.tiles[4 * (c.phase_wb_base[k] + b)];
            bool full;
            const int ev = tile_ev(b, full);
            int64_t e = b + 1;
            const bool seg_ok = td[2] != 0xFFFFFFFFu && ((td[3] >> 8) & 7u) < 6u && full && ev != -999 &&
                                (td[3] & 0xFFu) > 0;
            if (seg_ok) {
                while (e < nt) {
                    const uint32_t *te = &c.tiles[4 * (c.phase_wb_base[k] + e)];
                    bool f2;
                    if (te[2] != td[2] || te[3] != td[3] || te[1] != td[1] || tile_ev(e, f2) != ev || !f2) break;
                    e++;
                }
            }
            if (e - b >= SEG_MIN && seg_ok) {
                Compiled::Segment sg;
                sg.phase = k; sg.pos0 = c.phase_start[k] + 64 * b; sg.ntiles = (int32_t)(e - b);
                sg.adj_off = td[0]; sg.prog = td[2]; sg.nslots = td[3] & 0xFFu; sg.kind = (td[3] >> 8) & 7u;
                sg.ev = ev;
                sg.ztab = -1;
                if ((td[3] >> 11) & 1u) {                      // draw table of the program (shared)
                    auto zi = ztab_of.find(sg.prog);
                    if (zi == ztab_of.end() && c.nztab + ((int64_t)1 << sg.nslots) <= ((int64_t)1 << 20)) {
                        zi = ztab_of.emplace(sg.prog, c.nztab).first;
                        c.zprogs.push_back({sg.prog, sg.nslots, (uint32_t)c.nztab, 0u});
                        c.nztab += (int64_t)1 << sg.nslots;
                    }
                    if (zi != ztab_of.end()) sg.ztab = zi->second;
                }
                c.segments.push_back(sg);
            } else if (td[2] == 0xFFFFFFFFu || ((td[3] >> 8) & 7u) != 6u) {      // general tiles: own kernel
                for (int64_t t = b; t < e; t++) c.rest_tiles.push_back((uint32_t)t);
            }
            b = e;
        }
        c.phase_rest_base[k + 1] = (int64_t)c.rest_tiles.size();
    }
    if (c.rest_tiles.empty()) c.rest_tiles.push_back(0);
    if (getenv("NSK_VERBOSE")) {                 // layout report: tiles by kind, per colour
        for (int32_t k = 0; k < ncolors; k++) {
            int64_t kinds[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t b = 0; b < c.phase_wb_base[k + 1] - c.phase_wb_base[k]; b++) {
                const uint32_t *td = &c.tiles[4 * (c.phase_wb_base[k] + b)];
                kinds[td[2] == 0xFFFFFFFFu ? 8 : (td[3] >> 8) & 7u]++;
            }
            if (getenv("NSK_DEBUG_TILES"))
                for (int64_t b = 0, shown = 0; b < c.phase_wb_base[k + 1] - c.phase_wb_base[k] && shown < 3; b++) {
                    const uint32_t *td = &c.tiles[4 * (c.phase_wb_base[k] + b)];
                    if (td[2] != 0xFFFFFFFFu) continue;
                    shown++;
                    fprintf(stderr, "  per-lane tile %lld (gen tiles start %lld):", (long long)b, (long long)c.phase_gen_tile[k]);
                    for (int64_t p = c.phase_start[k] + 64 * b; p < c.phase_start[k] + 64 * b + 64; p += 9) {
                        const int64_t v = c.p_vid[p];
                        if (v < 0) { fprintf(stderr, " pad"); continue; }
                        std::vector<uint32_t> ww;
                        lane_words(v, ww);
                        fprintf(stderr, " v%lld f%d ev%d [", (long long)v, (int)fast[v], (int)d->variable[v].isEvidence);
                        for (size_t j = 0; j < ww.size(); j += 1 + ((ww[j] >> 24) & 7u))
                            fprintf(stderr, "%u:%u:%u ", ww[j] >> 27, (ww[j] >> 24) & 7u, ww[j] & 0xFFFFFFu);
                        fprintf(stderr, "]");
                    }
                    fprintf(stderr, "\n");
                }
            fprintf(stderr, "[nsk] colour %d: %lld positions, tiles uniform %lld pair %lld general %lld shape %lld "
                            "per-lane %lld; generic %lld (hub-style %lld)\n", (int)k,
                    (long long)(c.phase_start[k + 1] - c.phase_start[k]),
                    (long long)kinds[0], (long long)(kinds[2] + kinds[3] + kinds[4]), (long long)kinds[6],
                    (long long)kinds[7], (long long)kinds[8],
                    (long long)(c.phase_start[k + 1] - c.phase_fast_end[k]),
                    (long long)(c.phase_heavy_end[k] - c.phase_fast_end[k]));
        }
    }
    if (const char *dv = diag_env("NSK_DEBUG_VAR")) {       // (diagnostic: where a variable landed)
        for (const char *q = dv; *q;) {
            const int64_t v = atoll(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
            if (v < 0 || v >= nvar || c.color[v] < 0) continue;
            const int64_t p = c.iid[v];
            const int32_t k = c.color[v];
            const int64_t b = (p - c.phase_start[k]) / 64;
            const uint32_t *td = &c.tiles[4 * (c.phase_wb_base[k] + b)];
            fprintf(stderr, "[nsk] var %lld: colour %d position %lld tile %lld td {%u, %u, %u, %#x} kind %u slots %u",
                    (long long)v, (int)k, (long long)p, (long long)b, td[0], td[1], td[2], td[3], (td[3] >> 8) & 7u, td[3] & 0xFFu);
            for (const Compiled::Segment &sg : c.segments)
                if (p >= sg.pos0 && p < sg.pos0 + 64 * (int64_t)sg.ntiles)
                    fprintf(stderr, " | segment pos0 %lld ntiles %d prog %u nslots %u kind %u ev %d ztab %lld", (long long)sg.pos0,
                            sg.ntiles, sg.prog, sg.nslots, sg.kind, sg.ev, (long long)sg.ztab);
            if (td[2] != 0xFFFFFFFFu && ((td[3] >> 8) & 7u) < 6u) {
                fprintf(stderr, " | program:");
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t w_ = c.tile_hdr[td[2] + j];
                    fprintf(stderr, " [w%u c%u F%u cl%u ig%u fx%u]", w_ & 0xFFFFFFu, (w_ >> 24) & 7u, (w_ >> 27) & 1u, (w_ >> 28) & 1u,
                            (w_ >> 29) & 1u, (w_ >> 30) & 1u);
                }
            }
            fprintf(stderr, "\n");
        }
    }
    return NSK_OK;
}

// Pass 2 over the tiles: the lanes' words into the stream `adj` (total4 = its size in 16-byte units), chunk-major
// per tile.  general_words / lane_words are the compiler's per-variable word lists (general tiles / fast path).
template <typename WordsFn, typename LaneWordsFn>
static void fill_tiles(const nsk_graph_desc *d, Compiled &c, int64_t nwb, uint64_t total4, const std::vector<int32_t> &tile_colour,
                       WordsFn &&general_words, LaneWordsFn &&lane_words) {
    (void)d;
    // pass 2: fill the tiles.  Padding: member slots read the always-zero id (c.zero_id) in uniform
    // tiles, 0xFFFFFFFF in tiles with per-lane headers.
    c.adj.assign((size_t)total4 * 4 + 4, 0xFFFFFFFFu);
    std::vector<int64_t> nfast_part((size_t)compile_threads() + 1, 0);
    parallel_for(nwb, [&](int64_t tb0, int64_t tb1, int tix) {
        std::vector<uint32_t> words;
        int64_t nfast_here = 0;
        for (int64_t t = tb0; t < tb1; t++) {
            const int32_t k = tile_colour[t];
            const int64_t b = t - c.phase_wb_base[k];
            const int64_t p0 = c.phase_start[k] + 64 * b, p1 = std::min(p0 + 64, c.phase_fast_end[k]);
            const uint32_t *td = &c.tiles[4 * t];
            const uint64_t base = (uint64_t)td[0] * 4;
            const bool uniform = td[2] != 0xFFFFFFFFu && ((td[3] >> 8) & 7u) < 6u;
            const bool general = td[2] != 0xFFFFFFFFu && ((td[3] >> 8) & 7u) == 6u;
            const bool shape = td[2] != 0xFFFFFFFFu && ((td[3] >> 8) & 7u) == 7u;
            if (td[2] != 0xFFFFFFFFu) {     // padding: uniform tiles read the always-zero id, shape tiles variable / weight 0
                const uint32_t padw = uniform ? (uint32_t)c.zero_id : 0u;
                for (uint64_t j = 0; j < (uint64_t)td[1] * 64; j++) c.adj[base + j] = padw;
            }
            for (int64_t p = p0; p < p1; p++) {
                if (c.p_vid[p] < 0) continue;
                size_t out = 0;
                auto put = [&](uint32_t word) {
                    c.adj[base + 256 * (out / 4) + 4 * (uint64_t)(p - p0) + (out % 4)] = word;
                    out++;
                };
                if (general && c.phase_ep[k]) { nfast_here++; continue; }     // laid out by groups, below
                if (general) {               // entries padded to M member slots, then E entries
                    general_words(c.p_vid[p], &words);
                    const uint32_t M = (td[3] >> 16) & 7u, E = (td[3] & 0xFFu) / (2 + M);
                    uint32_t ne = 0;
                    for (size_t j = 0; j < words.size(); ne++) {
                        const uint32_t no = (words[j + 1] >> 4) & 7u;
                        put(words[j]); put(words[j + 1]);
                        for (uint32_t m = 2; m < 2 + no; m++)        // member: internal id | deo << 27
                            put((uint32_t)c.iid[words[j + m] & NSK_GEN_NULL] | (words[j + m] & ~NSK_GEN_NULL));
                        for (uint32_t m = no; m < M; m++) put(NSK_GEN_NULL);
                        j += 2 + no;
                    }
                    for (; ne < E; ne++) {                        // an entry no candidate value owns
                        put(0u);
                        put(14u << 14);
                        for (uint32_t m = 0; m < M; m++) put(NSK_GEN_NULL);
                    }
                    nfast_here++;
                    continue;
                }
                lane_words(c.p_vid[p], words);
                for (size_t j = 0; j < words.size();) {
                    const uint32_t nother = (words[j] >> 24) & 7u;
                    if (!uniform) put(words[j]);
                    else if (nother == 0) put((uint32_t)c.zero_id);      // the ignored slot of a member-less entry
                    for (uint32_t m = 1; m <= nother; m++) put((uint32_t)c.iid[words[j + m]]);
                    // shape tile: the member slots of the tile's layout that this lane's entry lacks
                    while (shape && out < (size_t)td[1] && (c.tile_hdr[td[2] + out] & 0x80000010u) == 0x80000010u)
                        put(NSK_SHAPE_NULL);
                    j += 1 + nother;
                }
                nfast_here++;
            }
        }
        nfast_part[tix] = nfast_here;
    }, 64);
    for (int64_t x : nfast_part) c.nfast += x;
}

// The generic path's index (slots, inverted index, CSR byte model), the inline generic stream, the gradient format
// and the census of what one sweep must move in the compiled layout (alg_bytes_*, layout_bytes_*).
template <typename ReadFn, typename LapFn>
static int build_index_and_census(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, bool head_by_vid,
                                  ReadFn &&for_each_read, LapFn &&lap, std::string &err) {
    const int64_t nvar = c.nvar, nfac = c.nfactor, nedge = c.nedge, nw = c.nweight;
    const int64_t LIM = (int64_t)1 << 31;
    (void)nvar; (void)nfac; (void)nedge; (void)nw; (void)head_by_vid; (void)LIM;
    // per position: first slot and first list entry (exclusive prefix sums of the per-position counts)
    std::vector<int64_t> pos_si((size_t)c.npos + 1, 0), pos_li((size_t)c.npos + 1, 0);
    parallel_for(c.npos, [&](int64_t pb0, int64_t pb1, int) {
        for (int64_t p = pb0; p < pb1; p++) {
            if (c.p_vid[p] < 0) continue;
            const nsk_variable &var = d->variable[c.p_vid[p]];
            const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
            int64_t nl = 0;
            for (int64_t k = 0; k < nslots; k++) nl += d->vmap[var.vtf_offset + k].factor_index_length;
            pos_si[p + 1] = nslots;
            pos_li[p + 1] = nl;
        }
    });
    for (int64_t p = 0; p < c.npos; p++) { pos_si[p + 1] += pos_si[p]; pos_li[p + 1] += pos_li[p]; }
    const int64_t nslot = pos_si[c.npos], nlist = pos_li[c.npos];
    if (nslot >= LIM - 1 || nlist >= LIM - 1) {
        err = "inverted index too large for 32-bit device indices";
        return NSK_E_RANGE;
    }
    c.nslot = nslot;
    c.slot_off.resize(nslot + 1);
    c.fidx.resize(nlist);
    const int64_t s_i = 4, s_v = c.vbytes, s_c = 4;
    const bool big_w = nw * 8 > (4 << 20);
    std::vector<uint8_t> generic_pos((size_t)c.npos + 1, 0);
    for (int32_t k = 0; k < ncolors; k++)
        for (int64_t p = c.phase_fast_end[k]; p < c.phase_start[k + 1]; p++) generic_pos[p] = 1;
    // (all byte counts are integers far below 2^53: the partial sums add up exactly in any order)
    std::vector<double> part_bytes((size_t)(compile_threads() + 1) * 4, 0.0);
    parallel_for(c.npos, [&](int64_t pb0, int64_t pb1, int tix) {
        std::vector<int64_t> uni;
        double bytes_inf = 0, bytes_learn = 0, lay_inf = 0, lay_learn = 0;
        for (int64_t p = pb0; p < pb1; p++) {
            const int64_t v = c.p_vid[p];
            int64_t si = pos_si[p], li = pos_li[p];
            if (v < 0) { c.p_slot[p] = (int32_t)si; c.p_init[p] = -1; continue; }      // padding position (-1: the learning
                                                                                   // table kernel's validity test)
            const nsk_variable &var = d->variable[v];
            const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
            c.p_info[p] = ((uint32_t)var.cardinality << 9) | ((var.dataType != 0) ? 0x100u : 0u) |
                          (uint32_t)(uint8_t)var.isEvidence;
            c.p_slot[p] = (int32_t)si;
            c.p_cnt[p] = (int32_t)c.cstart[v];
            c.p_init[p] = var.isEvidence == 1 ? c.v_init[v] : 0;         // read by the evidence chain only (learning.py:61-62)
            uni.clear();
            for (int64_t k = 0; k < nslots; k++) {
                const nsk_vtf &vt = d->vmap[var.vtf_offset + k];
                c.slot_off[si++] = (int32_t)li;
                for (int64_t j = 0; j < vt.factor_index_length; j++) {
                    const int64_t f = d->factor_index[vt.factor_index_offset + j];
                    c.fidx[li++] = (int32_t)f;
                    uni.push_back(f);
                }
            }
            if (nslots > 1) {
                std::sort(uni.begin(), uni.end());
                uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
            }
            // algorithmic bytes of this update, SURVEY.md section 8(d)
            double bi = 2 + s_i + s_v, bl = 0;
            for (int64_t f : uni) {
                const nsk_factor &fa = d->factor[f];
                const double ar = (double)std::max<int64_t>(fa.arity, 0);
                bi += s_i + 10 + ar * s_i + (is_cat_function(fa.factorFunction) ? ar * s_i : 0) +
                      (ar - 1) * s_v + (big_w ? 8 : 0);
                bl += (ar - 1) * s_v + 8 + 1 +
                      ((big_w && fa.weightId >= 0 && fa.weightId < nw && !c.w_fixed[fa.weightId]) ? 16 : 0);
            }
            bytes_inf += bi + 2 * s_c;
            bytes_learn += bi + bl + s_v;
            if (generic_pos[p]) { lay_inf += bi + 2 * s_c; lay_learn += bi + bl + s_v; }
        }
        part_bytes[4 * tix] = bytes_inf; part_bytes[4 * tix + 1] = bytes_learn;
        part_bytes[4 * tix + 2] = lay_inf; part_bytes[4 * tix + 3] = lay_learn;
    });
    double bytes_inf = 0, bytes_learn = 0;
    double lay_inf = 0, lay_learn = 0;         // generic-path positions: the CSR model is their layout
    for (size_t t = 0; t * 4 < part_bytes.size(); t++) {
        bytes_inf += part_bytes[4 * t]; bytes_learn += part_bytes[4 * t + 1];
        lay_inf += part_bytes[4 * t + 2]; lay_learn += part_bytes[4 * t + 3];
    }
    c.slot_off[nslot] = (int32_t)nlist;

    lap("slots + CSR bytes");
    // ---- inline generic stream: for the positions handled by the one-lane generic kernels, every
    // factor record of every slot copied in list order, members included, so that a lane reads its
    // update sequentially instead of chasing fidx -> factor -> fmap through three arrays
    c.gs_off.assign((size_t)nslot + 1, 0);
    {
        auto members_stored = [&](const nsk_factor &fa) -> int64_t {   // edges the function may read
            const int fn = fa.factorFunction;
            int64_t need = (fn == 21 || fn == 22 || fn == 25 || fn == 26) ? 2 : (fn == 23 || fn == 24) ? 3
                         : (fn >= 18 && fn <= 20) ? 1 : (fn == 3 ? 1 : 0);
            int64_t n = std::max<int64_t>(std::max<int64_t>(fa.arity, 0), need);
            if (fn == 30 && fa.ftv_offset >= 0 && fa.ftv_offset < nedge)
                n = std::max<int64_t>(n, d->variable[d->fmap[fa.ftv_offset].vid].cardinality - 1);
            if (fn == -1) n = 0;
            return n;
        };
        uint64_t units = 2;                       // unit 0/1 unused so that offset 0 means "none"
        for (int32_t k = 0; k < ncolors; k++)
            for (int64_t p = c.phase_heavy_end[k]; p < c.phase_start[k + 1]; p++) {
                const int64_t v = c.p_vid[p];
                if (v < 0) continue;
                const nsk_variable &var = d->variable[v];
                const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
                for (int64_t kk = 0; kk < nslots; kk++) {
                    const nsk_vtf &vt = d->vmap[var.vtf_offset + kk];
                    for (int64_t j = 0; j < vt.factor_index_length; j++)
                        units += 4 + (uint64_t)members_stored(d->factor[d->factor_index[vt.factor_index_offset + j]]);
                }
            }
        if (units >= ((uint64_t)1 << 32)) { err = "inline generic stream too large"; return NSK_E_RANGE; }
        c.gstream.assign((size_t)units * 2, 0);
        uint64_t at = 2;
        for (int32_t k = 0; k < ncolors; k++)
            for (int64_t p = c.phase_heavy_end[k]; p < c.phase_start[k + 1]; p++) {
                const int64_t v = c.p_vid[p];
                if (v < 0) continue;
                const nsk_variable &var = d->variable[v];
                const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
                for (int64_t kk = 0; kk < nslots; kk++) {
                    const nsk_vtf &vt = d->vmap[var.vtf_offset + kk];
                    c.gs_off[c.p_slot[p] + kk] = (uint32_t)at;
                    for (int64_t j = 0; j < vt.factor_index_length; j++) {
                        const int64_t f = d->factor_index[vt.factor_index_offset + j];
                        const nsk_factor &fa = d->factor[f];
                        const int64_t nm = members_stored(fa);
                        uint32_t *u = &c.gstream[at * 2];
                        u[0] = c.f_rec[4 * f]; u[1] = c.f_rec[4 * f + 2];              // head, weightId
                        u[2] = c.f_rec[4 * f + 1]; u[3] = (uint32_t)f;                 // ftv_offset, factor id
                        memcpy(&u[4], &fa.featureValue, 8);
                        u[6] = (uint32_t)nm; u[7] = 0;
                        for (int64_t m = 0; m < nm; m++) {
                            const int64_t l = fa.ftv_offset + m;
                            u[8 + 2 * m] = (uint32_t)c.m_rec[2 * l];
                            u[9 + 2 * m] = (uint32_t)c.m_rec[2 * l + 1];
                        }
                        at += 4 + (uint64_t)nm;
                    }
                }
            }
    }
    lap("generic stream");
    choose_gradient_format(d, c);
    c.alg_bytes_inference = bytes_inf;
    c.alg_bytes_learning = bytes_learn;
    lap("packed-gradient check");
    // ---- layout bytes: what one sweep must move in the compiled layout.  Tile words (padding
    // included), position arrays, the distinct neighbour values a colour class reads, the stores and
    // the tally read-modify-write; materialised weight rows / gathered weights when the table
    // exceeds the L2; generic-path positions as in the CSR model above.
    {
        std::vector<uint64_t> seen((size_t)(nvar + 63) / 64);           // bit b: the class reads variable b
        for (int32_t k = 0; k < ncolors; k++) {
            double words = 0, wrows = 0, ep_wt_bytes = 0;
            for (int64_t b = 0; b < c.phase_wb_base[k + 1] - c.phase_wb_base[k]; b++) {
                const uint32_t *td = &c.tiles[4 * (c.phase_wb_base[k] + b)];
                const uint32_t kind = td[2] == 0xFFFFFFFFu ? 8u : (td[3] >> 8) & 7u;
                words += (double)td[1] * 64 * 4;
                const bool seg_like = kind < 6u;                 // uniform tiles: p_vid + tally only
                lay_inf += 64.0 * (4 + (seg_like ? 0 : 4));
                lay_learn += 64.0 * (4 + 4 + s_v);               // p_vid, p_info, p_init
                if (big_w && kind == 6u) wrows += (double)((td[3] & 0xFFu) / (2 + ((td[3] >> 16) & 7u))) * 64 * 8;
                if (big_w && kind == 7u) wrows += (double)(td[3] & 0xFFu) * 64 * 8 / 2;   // ~ one header per two words
            }
            if (c.phase_ep[k])                         // entry-parallel groups: their rows instead of tile words;
                for (int64_t gi = c.phase_ep_base[k]; gi < c.phase_ep_base[k + 1]; gi++) {
                    const uint32_t *gd = &c.ep_desc[(size_t)gi * 4];
                    double sr = 0;
                    double rows = 0;
                    for (uint32_t m = 0; m < 4; m++) {
                        const double r = (double)((gd[1] >> (8 * m)) & 255u) + (double)((gd[3] >> (8 * m)) & 255u);
                        sr += r * (2 + m);
                        rows += r;
                    }
                    words += sr * 256;
                    ep_wt_bytes += rows * 64 * 8;                      // inference: the materialised weight of every entry
                    if (big_w) wrows += rows * 64 * 8;                 // learning: one gathered weight per entry
                }
            lay_inf += words + (c.phase_ep[k] ? ep_wt_bytes : wrows);
            lay_learn += words + 3 * wrows;            // weight gathers + one 16-byte atomic per visit
            int64_t distinct = 0, nfastpos = 0, ncatpos = 0;
            std::fill(seen.begin(), seen.end(), 0ull);
            {   // the class's positions over the host threads; a neighbour counts for the thread that sets its bit
                const int64_t pa = c.phase_start[k], pb = c.phase_fast_end[k];
                std::vector<int64_t> part((size_t)compile_threads() * 3, 0);
                parallel_for(pb - pa, [&](int64_t b0, int64_t b1, int t) {
                    int64_t dn = 0, nf = 0, nc = 0;
                    for (int64_t p = pa + b0; p < pa + b1; p++) {
                        const int64_t v = c.p_vid[p];
                        if (v < 0) continue;
                        if (d->variable[v].cardinality == 2) nf++; else nc++;
                        for_each_read(v, [&](int64_t b) {
                            if (b == v) return;
                            const uint64_t bit = 1ull << (b & 63);
                            if (!(__atomic_fetch_or(&seen[(size_t)b >> 6], bit, __ATOMIC_RELAXED) & bit)) dn++;
                        });
                    }
                    part[(size_t)t * 3] = dn; part[(size_t)t * 3 + 1] = nf; part[(size_t)t * 3 + 2] = nc;
                });
                for (size_t t = 0; t < part.size(); t += 3) { distinct += part[t]; nfastpos += part[t + 1]; ncatpos += part[t + 2]; }
            }
            lay_inf += (double)distinct * s_v + (double)(nfastpos + ncatpos) * s_v + 2.0 * nfastpos + 2.0 * s_c * ncatpos;
            lay_learn += 2.0 * distinct * s_v + 2.0 * (nfastpos + ncatpos) * s_v;
        }
        for (const Compiled::Segment &sg : c.segments) {    // inference over table segments: a tile with implicit
            if (sg.aff < 0) continue;                        // adjacency reads 16 bytes per chunk, not 64 x 16
            const int nch = sg.nslots > 4 ? 2 : 1;
            for (int64_t t = 0; t < sg.ntiles; t++)
                if (c.seg_aff[((size_t)sg.aff + (size_t)t * nch) * 4] != 0xFFFFFFFFu) {
                    lay_inf -= (double)nch * (64 * 16 - 16);
                    lay_learn -= (double)nch * (64 * 16 - 16);
                }
        }
        for (const Compiled::Segment &sg : c.segments)      // the table kernels key their generators by
            if (sg.ztab >= 0) {                              // position: no p_vid read; learning: no p_info either
                lay_inf -= (double)sg.ntiles * 64 * 4;
                lay_learn -= (double)sg.ntiles * 64 * 8;
            }
        c.layout_bytes_inference = lay_inf;
        c.layout_bytes_learning = lay_learn;
    }
    lap("layout bytes");
    return NSK_OK;
}

// Affine runs of one exact class in a first layout (place_variables): the class holds `count` variables at positions
// [start, start + count), id order, all with the same slot program.  B_j(r) = (position of member j of the r-th
// variable) - r is constant along a run.  A position whose members disagree with the run's bases in some slot is an
// EXCEPTION when its successors agree again (the end cell of a grid row: its neighbour lives in the border class), a
// BREAK when they settle on other bases (the next grid row).  Out: the empty positions to put in front of ranks so
// that every long run starts on a multiple of 256 (the class itself will start on one); false: nothing worth padding.
static bool find_run_padding(const nsk_graph_desc *d, const Compiled &c, int64_t start, int64_t count,
                             std::vector<std::pair<int64_t, int64_t>> &at, int64_t &total) {
    at.clear();
    total = 0;
    // member slots of the class (its variables share one program: same factor functions and member counts)
    int ns = 0;
    {
        const int64_t v = c.p_vid[start];
        const nsk_vtf &vt = d->vmap[d->variable[v].vtf_offset];
        for (int64_t j = 0; j < vt.factor_index_length; j++) {
            const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
            if (fa.factorFunction == -1) continue;
            for (int64_t l = fa.ftv_offset; l < fa.ftv_offset + fa.arity; l++) if (d->fmap[l].vid != v) ns++;
        }
    }
    if (ns == 0 || ns > 8) return false;
    // provisional ids: a sampled variable's position, the ghosts this handle reads behind them in id order
    std::vector<int64_t> B((size_t)count * (size_t)ns);
    const int64_t NONE = INT64_MIN / 2;
    parallel_for(count, [&](int64_t rb0, int64_t rb1, int) {
        for (int64_t r = rb0; r < rb1; r++) {
            const int64_t v = c.p_vid[start + r];
            const nsk_vtf &vt = d->vmap[d->variable[v].vtf_offset];
            int s = 0;
            for (int64_t j = 0; j < vt.factor_index_length && s <= ns; j++) {
                const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
                if (fa.factorFunction == -1) continue;
                for (int64_t l = fa.ftv_offset; l < fa.ftv_offset + fa.arity; l++) {
                    const int64_t m = d->fmap[l].vid;
                    if (m == v) continue;
                    int64_t id = c.v_pos[m];
                    if (id < 0) {
                        const auto it = std::lower_bound(c.ghost_needs.begin(), c.ghost_needs.end(), (int32_t)m);
                        id = (it != c.ghost_needs.end() && *it == m) ? c.npos + (it - c.ghost_needs.begin()) : NONE;
                    }
                    if (s < ns) B[(size_t)r * ns + s] = id == NONE ? NONE + r : id - r;     // (NONE + r: equal to nothing)
                    s++;
                }
            }
            for (; s < ns; s++) B[(size_t)r * ns + s] = NONE + r;
        }
    });
    auto miss = [&](int64_t r, const int64_t *base) { int m = 0; for (int j = 0; j < ns; j++) m += B[(size_t)r * ns + j] != base[j]; return m; };
    // per-slot mode over a short window ahead of r
    auto settle = [&](int64_t r, int64_t *out) {
        for (int j = 0; j < ns; j++) {
            int64_t best = B[(size_t)r * ns + j];
            int bestn = 0;
            for (int64_t a = r; a < std::min(count, r + 5); a++) {
                int n = 0;
                for (int64_t b2 = r; b2 < std::min(count, r + 5); b2++) n += B[(size_t)b2 * ns + j] == B[(size_t)a * ns + j];
                if (n > bestn) { bestn = n; best = B[(size_t)a * ns + j]; }
            }
            out[j] = best;
        }
    };
    std::vector<int64_t> run_start;          // ranks
    int64_t cur[8], nxt[8];
    settle(0, cur);
    run_start.push_back(0);
    for (int64_t r = 1; r < count; r++) {
        if (miss(r, cur) == 0) continue;
        settle(r, nxt);
        bool same = true;
        for (int j = 0; j < ns; j++) same = same && nxt[j] == cur[j];
        if (same) continue;                                           // an exception: its successors agree with the run
        // a break: the new run starts at the first position that fits the new bases better than the old ones
        int64_t rb = r;
        while (rb < std::min(count, r + 4) && miss(rb, nxt) >= miss(rb, cur)) rb++;
        if (rb >= std::min(count, r + 4)) rb = r;
        if (rb > run_start.back()) run_start.push_back(rb);
        for (int j = 0; j < ns; j++) cur[j] = nxt[j];
        r = rb;
    }
    run_start.push_back(count);
    // long runs start on multiples of 256 when that wastes little
    int64_t posn = 0;                         // position relative to the class start (a multiple of 256)
    int64_t padded = 0, covered = 0;
    for (size_t i = 0; i + 1 < run_start.size(); i++) {
        const int64_t len = run_start[i + 1] - run_start[i];
        const int64_t waste = (256 - len % 256) % 256;
        const bool good = len >= 384 && waste * 12 <= len;
        if (good && posn % 256 != 0) {
            const int64_t pad = 256 - posn % 256;
            at.push_back({run_start[i], pad});
            total += pad;
            posn += pad;
        }
        if (good) { padded++; covered += len; }
        posn += len;
    }
    if (getenv("NSK_DEBUG_WIDE")) {
        fprintf(stderr, "[nsk] class at %lld (%lld variables, %d slots): %zu runs, %lld padded, %lld empty positions; runs start at", (long long)start,
                (long long)count, ns, run_start.size() - 1, (long long)padded, (long long)total);
        for (size_t i = 0; i + 1 < run_start.size() && i < 12; i++) fprintf(stderr, " %lld", (long long)run_start[i]);
        fprintf(stderr, "\n");
    }
    if (padded == 0 || total * 10 > count || covered * 2 < count) { at.clear(); total = 0; return false; }
    return true;
}

// Positions: colour-major.  Inside a colour the fast variables grouped by class -- exact program, exact shape, padded
// shape (per id range) -- so that the 64 lanes of a tile share one slot program / word layout; every class with at
// least 64 members starts on a tile boundary (the gap is padded with empty positions, p_vid = -1); smaller classes
// share a tail in id order; then the general tiles' variables (sorted), then the generic-path variables (binned by
// work).  fast[v]: 1 fast path, 2 general tile, 0 generic (variables that fit no class move from 1 to 2 or 0 here).
// [shape_at[k], shape_end[k]) = the positions of colour k's shape classes.
template <typename WordsFn, typename LapFn>
static int place_variables(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, std::vector<uint8_t> &fast,
                           WordsFn &&general_words, bool no_general, int64_t shape_words, int64_t shape_words_ep,
                           std::vector<int64_t> &shape_at, std::vector<int64_t> &shape_end, LapFn &&lap, std::string &err) {
    const int64_t nvar = c.nvar, nfac = c.nfactor, nedge = c.nedge, nw = c.nweight;
    const int64_t LIM = (int64_t)1 << 31;
    (void)nfac; (void)nedge; (void)nw; (void)LIM; (void)err;
    // id blocks of the general tiles' sort order (per-lane walk / entry-parallel groups)
    const int64_t gen_block = diag_env("NSK_GEN_BLOCK") ? std::max<int64_t>(64, atoll(diag_env("NSK_GEN_BLOCK"))) : 262144;
    const int64_t ep_block = diag_env("NSK_EP_BLOCK") ? std::max<int64_t>(64, atoll(diag_env("NSK_EP_BLOCK"))) : 1024;
    // sig: exact program (function, member count, weight id per entry, evidence flag);
    // shp: shape only (member count per entry, evidence flag), 0 when the stream would exceed
    //      16 words.  General-tile variables are not classed: they are sorted (below).
    // pshp: the shape with every entry's member count rounded up to even ("padded" shape): variables whose
    //      exact shape is rare share tiles with near shapes, the missing member slots filled with null
    //      words (NSK_SHAPE_NULL) -- with individual weights and lists of 7+ entries the exact shapes
    //      (2^(entries-1) of them on the weighted boolean graph) no longer fill tiles
    std::vector<uint64_t> sig, shp, pshp;           // (sized below, when the graph has such variables at all)
    // Shape classes are formed per id range ("part") of the graph: the lanes of a shape tile then come from
    // one part, and the values and weights they gather -- mostly those of id neighbours -- from a
    // correspondingly narrow stretch of every colour's positions (the kernels hand an XCD a contiguous
    // eighth of the colour's tiles, so one L2 serves those gathers).  One class over the whole id range
    // put 64 unrelated variables into a tile: the learning sweep of the 4M-variable weighted boolean
    // graph missed the L2 11 times per variable.  Parts of >= 2^18 ids keep the leftovers (< 64 members
    // of a shape in a part, general tiles) few: that graph's learning sweep, 8 / 16 / 32 parts: 3.71 / 4.07 /
    // 4.06e9 updates/s (tools/sessions/history/r4_s26.sh).
    const bool no_pshape = diag_env("NSK_NO_PAD_SHAPE") != nullptr || diag_env("NSK_NO_SHAPE") != nullptr;
    const int64_t shape_parts = diag_env("NSK_SHAPE_PARTS") ? std::max<int64_t>(1, atoll(diag_env("NSK_SHAPE_PARTS")))
                                                            : std::max<int64_t>(1, std::min<int64_t>(64, (nvar + (1 << 18) - 1) >> 18));
    std::vector<int64_t> nfast_of((size_t)ncolors, 0), ngen_of((size_t)ncolors, 0), ngt_of((size_t)ncolors, 0);
    for (int64_t v = 0; v < nvar; v++) {
        if (c.color[v] < 0) continue;
        if (fast[v] == 2) ngt_of[c.color[v]]++;
        else if (!fast[v]) ngen_of[c.color[v]]++;
        else nfast_of[c.color[v]]++;
    }
    {
        int64_t nclassed = 0;
        for (int32_t k = 0; k < ncolors; k++) nclassed += nfast_of[k];
        if (nclassed > 0) { sig.assign((size_t)nvar, 0); shp.assign((size_t)nvar, 0); pshp.assign((size_t)nvar, 0); }
    }
    parallel_for(nvar, [&](int64_t vb0, int64_t vb1, int) {
    for (int64_t v = vb0; v < vb1; v++) {
        if (c.color[v] < 0 || fast[v] != 1) continue;
        const nsk_variable &var = d->variable[v];
        const nsk_vtf &vt = d->vmap[var.vtf_offset];
        // (the evidence flag is multiplied in before the first word: a plain xor would cancel
        // against the low bit of the first weight id / member count)
        uint64_t h = (0xcbf29ce484222325ull ^ (uint64_t)(uint8_t)var.isEvidence) * 0x100000001b3ull;
        uint64_t h2 = (h ^ 0x9e3779b97f4a7c15ull ^ ((uint64_t)(v * shape_parts / std::max<int64_t>(nvar, 1)) << 40)) * 0x100000001b3ull;
        uint64_t h3 = (h2 ^ 0xd6e8feb86659fd93ull) * 0x100000001b3ull;
        int64_t nwords = 0, pwords = 0;
        uint64_t maxo = 0;
        for (int64_t j = 0; j < vt.factor_index_length; j++) {
            const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
            uint64_t others = 0;
            if (fa.factorFunction != -1)
                for (int64_t l = fa.ftv_offset; l < fa.ftv_offset + fa.arity; l++)
                    if (d->fmap[l].vid != v) others++;
            const uint64_t word = ((uint64_t)(fa.factorFunction + 1) << 27) | (others << 24) | (uint64_t)fa.weightId;
            h = (h ^ word) * 0x100000001b3ull;
            h ^= h >> 29;
            h2 = (h2 ^ (others + 1)) * 0x100000001b3ull;
            h2 ^= h2 >> 31;
            const uint64_t padded = (others + 1) & ~(uint64_t)1;
            maxo = std::max(maxo, others);
            h3 = (h3 ^ (padded + 1)) * 0x100000001b3ull;
            h3 ^= h3 >> 31;
            nwords += 1 + (int64_t)others;
            pwords += 1 + (int64_t)padded;
        }
        sig[v] = h | 1;
        // (a variable the entry-parallel groups can take -- <= 3 other members per entry, <= 16 entries -- joins a
        //  shape class only with a list of a few words, shape_words_ep)
        const int64_t lim = (maxo <= 3 && vt.factor_index_length <= 16) ? shape_words_ep : shape_words;
        shp[v] = nwords <= lim ? (h2 | 1) : 0;
        pshp[v] = (pwords <= lim && !no_pshape) ? (h3 | 1) : 0;
    }
    });
    lap("positions: signatures");
    // (hash maps: with one weight per factor every variable is a class of its own -- millions of keys;
    //  nothing below depends on their iteration order.  The colours are independent: one thread each.)
    typedef std::unordered_map<uint64_t, std::pair<int64_t, int64_t>> ClassMap;        // key -> (count, first vid)
    std::vector<ClassMap> classes((size_t)ncolors), shapes((size_t)ncolors), pshapes((size_t)ncolors);
    // a class gets tiles of its own when it fills at least one (64 members) -- or whatever its
    // size when the colour has only a few small classes (then padding them costs nothing
    // and no tile is left with mixed programs, e.g. the corner cells of a grid)
    std::vector<int64_t> min_class((size_t)ncolors, 64);
    parallel_for(ncolors, [&](int64_t kb0, int64_t kb1, int) {
    for (int32_t k = (int32_t)kb0; k < (int32_t)kb1; k++) {
        if (nfast_of[k] == 0) continue;                 // (nothing to class: four scans of the variables saved)
        ClassMap &cls = classes[k], &shs = shapes[k], &pss = pshapes[k];
        cls.reserve((size_t)nfast_of[k]);
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] != k || fast[v] != 1) continue;
            auto &e = cls[sig[v]];
            if (e.first++ == 0) e.second = v;
        }
        int64_t nsmall = 0;
        for (auto &kv : cls) if (kv.second.first < 64) nsmall++;
        if (nsmall <= 16) min_class[k] = 1;
        // variables outside the big exact classes are grouped by shape
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] != k || fast[v] != 1 || shp[v] == 0 || cls[sig[v]].first >= min_class[k]) continue;
            auto &e = shs[shp[v]];
            if (e.first++ == 0) e.second = v;
        }
        // ... and the ones whose exact shape fills no tile by padded shape
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] != k || fast[v] != 1 || pshp[v] == 0 || cls[sig[v]].first >= min_class[k]) continue;
            if (shp[v] != 0 && shs[shp[v]].first >= 64) continue;
            auto &e = pss[pshp[v]];
            if (e.first++ == 0) e.second = v;
        }
        // what neither an exact nor a shape class can take would end in mixed tiles with per-lane
        // parsing: the general tiles' sorted layout serves those variables better
        for (int64_t v = 0; v < nvar && !no_general; v++) {
            if (c.color[v] != k || fast[v] != 1 || cls[sig[v]].first >= min_class[k]) continue;
            if (shp[v] != 0 && shs[shp[v]].first >= 64) continue;
            if (pshp[v] != 0 && pss[pshp[v]].first >= 64) continue;
            if (shp[v] != 0) shs[shp[v]].first--;
            if (pshp[v] != 0) pss[pshp[v]].first--;
            nfast_of[k]--;
            if (general_words(v, nullptr)) { fast[v] = 2; ngt_of[k]++; }
            else { fast[v] = 0; ngen_of[k]++; }          // long lists: wave-per-variable / generic kernels
        }
    }
    }, 1);
    // a colour whose exact / shape classes are a sliver next to its general tiles gives them up:
    // their few tiles would cost two or three extra launches per class and sweep
    for (int32_t k = 0; k < ncolors && !no_general; k++) {
        if (nfast_of[k] == 0 || nfast_of[k] * 20 >= ngt_of[k]) continue;
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] != k || fast[v] != 1) continue;
            nfast_of[k]--;
            if (general_words(v, nullptr)) { fast[v] = 2; ngt_of[k]++; }
            else { fast[v] = 0; ngen_of[k]++; }
        }
        classes[k].clear();
        shapes[k].clear();
        pshapes[k].clear();
    }
    if (getenv("NSK_VERBOSE")) { int64_t ng = 0, nf = 0; for (int32_t k = 0; k < ncolors; k++) { ng += ngt_of[k]; nf += nfast_of[k]; } fprintf(stderr, "[nsk] after the classes: %lld general-tile variables, %lld classed\n", (long long)ng, (long long)nf); }
    lap("positions: classes");
    // ---- what does not depend on the positions: work bins of the generic-path variables, the general tiles' order ----
    // generic-path variables of a colour are ordered by the work of one update (factor-list
    // lengths x arities over all candidate values, binned) so that the 64 lanes of a wave finish
    // together; inside a bin: variable id.
    std::vector<uint32_t> gw;
    std::vector<uint8_t> work_bin(nvar, 0);
    for (int64_t v = 0; v < nvar; v++) {
        if (c.color[v] < 0 || fast[v]) continue;
        const nsk_variable &var = d->variable[v];
        const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
        int64_t work = 0, listlen = 0;
        for (int64_t kk = 0; kk < nslots; kk++) {
            const nsk_vtf &vt = d->vmap[var.vtf_offset + kk];
            listlen += vt.factor_index_length;
            for (int64_t j = 0; j < vt.factor_index_length; j++)
                work += 2 + std::max<int64_t>(d->factor[d->factor_index[vt.factor_index_offset + j]].arity, 0);
        }
        if (var.dataType == 0) work *= var.cardinality;
        int bin = 0;
        while (work > 8 && bin < 39) { work = work * 3 / 4; bin++; }     // ~log_{4/3} bins
        work_bin[v] = (uint8_t)(40 - bin);                               // heavier variables first
        // hubs: a whole wave works on one such variable (heavy_update in k_gibbs_general / k_learn_heavy)
        if (listlen >= NSK_HEAVY_LIST && !diag_env("NSK_NO_HEAVY")) work_bin[v] = 0;
    }
    // a colour with few generic-path variables gives every one of them a wave: the one-lane
    // kernel's run time is the latency of its longest serial walk however few lanes are busy
    if (!diag_env("NSK_NO_HEAVY"))
        for (int64_t v = 0; v < nvar; v++)
            if (c.color[v] >= 0 && !fast[v] && ngen_of[c.color[v]] <= NSK_FEW_GENERIC) work_bin[v] = 0;
    std::vector<std::vector<int64_t>> bin_count((size_t)ncolors, std::vector<int64_t>(42, 0));
    for (int64_t v = 0; v < nvar; v++)
        if (c.color[v] >= 0 && !fast[v]) bin_count[c.color[v]][work_bin[v] + 1]++;
    lap("positions: work bins");
    // general-tile variables of a colour: sorted by (entries, most other members of an entry),
    // largest first, and cut into tiles of 64 -- a tile's layout is the maximum over its lanes,
    // so neighbours in this order waste the least padding (SELL-C-sigma)
    std::vector<std::vector<std::pair<int64_t, int64_t>>> order((size_t)ncolors);   // (key, vid)
    std::vector<uint8_t> g_ne(nvar, 0), g_mo(nvar, 0);      // entries / widest entry of a general lane
    parallel_for(nvar, [&](int64_t vb0, int64_t vb1, int) {
        std::vector<uint32_t> w;
        for (int64_t v = vb0; v < vb1; v++) {
            if (c.color[v] < 0 || fast[v] != 2) continue;
            general_words(v, &w);
            int64_t ne = 0, mo = 0;
            for (size_t j = 0; j < w.size(); j += 2 + ((w[j + 1] >> 4) & 7u)) {
                ne++;
                mo = std::max<int64_t>(mo, (w[j + 1] >> 4) & 7u);
            }
            g_ne[v] = (uint8_t)ne; g_mo[v] = (uint8_t)mo;
        }
    });
    lap("positions: lane sizes");
    // entry-parallel groups (nsk_compile.h ep_desc) serve a colour whose general variables all
    // have entries of at most 3 other members and at most 16 entries (ordinal: 5 bits, LDS slots)
    c.phase_ep.assign((size_t)ncolors, 0);
    c.phase_ep_emax.assign((size_t)ncolors, 0);
    if (!diag_env("NSK_NO_EP") && nw < ((int64_t)1 << 27)) {
        for (int32_t k = 0; k < ncolors; k++) c.phase_ep[k] = ngt_of[k] > 0 ? 1 : 0;
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] < 0 || fast[v] != 2) continue;
            if (g_mo[v] > 3 || g_ne[v] > 16) c.phase_ep[c.color[v]] = 0;
            c.phase_ep_emax[c.color[v]] = std::max<int32_t>(c.phase_ep_emax[c.color[v]], g_ne[v]);
        }
    }
    // key: categorical lanes first (their tiles form a launch of their own), then blocks
    // of gen_block consecutive ids (sigma of SELL-C-sigma: each XCD walks a contiguous
    // run of tiles, so its L2 then sees one slice of the value array instead of all of
    // it), largest layouts first inside a block.  Entry-parallel groups carry no padding to the
    // widest lane, so their colours are cut into small id blocks -- a group's member values then
    // share cache lines --, with the variables of more than 8 entries (two LDS passes per group)
    // in front of the others.  One colour per thread: collect its variables, sort them.
    auto collect = [&](int32_t k) {
        std::vector<std::pair<int64_t, int64_t>> &ord = order[(size_t)k];
        ord.reserve((size_t)ngt_of[k]);
        const bool epk = c.phase_ep[k] != 0;
        const int64_t gb = epk ? ep_block : gen_block;
        for (int64_t v = 0; v < nvar; v++) {
            if (c.color[v] != k || fast[v] != 2) continue;
            const int64_t ne = g_ne[v], mo = g_mo[v];
            const int64_t catv = c.v_card[v] > 2 ? 0 : 1;
            const int64_t small = (epk && ne <= 8) ? 1 : 0;
            ord.push_back({(small << 51) | (catv << 50) | ((v / gb) << 20) | (0xFFFFF - (ne * 8 + mo)), v});
        }
        std::sort(ord.begin(), ord.end());
    };
    {
        std::vector<std::thread> sorters;             // (few colours only)
        for (int32_t k = 0; k < ncolors; k++) {
            if (ncolors <= 64 && compile_threads() > 1) sorters.emplace_back([&, k] { collect(k); });
            else collect(k);
        }
        for (auto &t : sorters) t.join();
    }
    lap("positions: lane order");

    // ---- positions.  Run padding (ClassPad): inside an exact class -- id order -- the members of consecutive variables
    // are, on regular graphs, consecutive positions of another class (a grid row's neighbours are the rows above,
    // below and beside it): an AFFINE RUN.  The table kernels take such positions four to a lane (nsk_compile.h
    // seg_wide) when a run starts on a multiple of 256, so the positions are laid out twice when that pays: the
    // first layout finds the runs (find_run_padding), the second starts every long run on a quad boundary, with
    // empty positions (p_vid = -1) in front.
    struct ClassPad { std::vector<std::pair<int64_t, int64_t>> at; int64_t total = 0; };     // (rank in the class, empty positions in front of it)
    struct ClassAt { int32_t k; uint64_t key; int64_t start, count; };
    struct Cursor { int64_t next, rank; size_t bi; const ClassPad *pad; };
    std::vector<std::unordered_map<uint64_t, ClassPad>> pads((size_t)ncolors);
    std::vector<ClassAt> exact_at;
    auto lay = [&]() -> int {
        std::vector<int64_t> next_gen((size_t)ncolors, 0), tail_at((size_t)ncolors, 0), gt_at((size_t)ncolors, 0);
        std::vector<std::vector<int64_t>> gen_bin_start;
        std::vector<std::unordered_map<uint64_t, Cursor>> start((size_t)ncolors);
        std::vector<std::map<uint64_t, int64_t>> start2((size_t)ncolors), start3((size_t)ncolors);
        exact_at.clear();
        c.nsampled = 0;
        int64_t pos = 0;
        for (int32_t k = 0; k < ncolors; k++) {
            pos = (pos + 127) / 128 * 128;      // tiles sit on multiples of 64, tile pairs on multiples of
            if (!pads[k].empty()) pos = (pos + 255) / 256 * 256;
            c.phase_start[k] = pos;             // 128: a lane's position & 63 is its lane (generator ids)
            int64_t nbig = 0;
            for (int level = 0; level < 3; level++) {
                if (level == 1) shape_at[k] = pos;
                ClassMap &cm = level == 0 ? classes[k] : level == 1 ? shapes[k] : pshapes[k];
                std::vector<std::pair<int64_t, uint64_t>> big;        // (first vid, key)
                const int64_t need = level == 0 ? min_class[k] : 64;
                for (auto &kv : cm)
                    if (kv.second.first >= need) { big.push_back({kv.second.second, kv.first}); nbig += kv.second.first; }
                std::sort(big.begin(), big.end());
                for (auto &bc : big) {
                    const int64_t count = cm[bc.second].first;
                    if (level == 0) {
                        const auto pit = pads[k].find(bc.second);
                        const ClassPad *pd = pit == pads[k].end() ? nullptr : &pit->second;
                        if (pd) pos = (pos + 255) / 256 * 256;          // a padded class owns whole quads
                        Cursor cu{pos, 0, 0, pd};
                        if (pd && !pd->at.empty() && pd->at[0].first == 0) { cu.next += pd->at[0].second; cu.bi = 1; }
                        start[k][bc.second] = cu;
                        exact_at.push_back(ClassAt{k, bc.second, pos, count});
                        pos += count + (pd ? pd->total : 0);
                        if (pd) pos = (pos + 255) / 256 * 256;
                    } else {
                        (level == 1 ? start2[k] : start3[k])[bc.second] = pos;
                        pos += count;
                    }
                    pos = c.phase_start[k] + (pos - c.phase_start[k] + 63) / 64 * 64;
                }
            }
            tail_at[k] = pos;
            shape_end[k] = pos;
            pos += nfast_of[k] - nbig;
            pos = c.phase_start[k] + (pos - c.phase_start[k] + 63) / 64 * 64;   // tiles own all 64 positions
            gt_at[k] = pos;                                                     // general tiles
            c.phase_gen_tile[k] = (pos - c.phase_start[k]) / 64;
            pos += ngt_of[k];
            pos = c.phase_start[k] + (pos - c.phase_start[k] + 63) / 64 * 64;
            c.phase_fast_end[k] = pos;
            next_gen[k] = pos;
            pos += ngen_of[k];
            c.phase_end[k] = pos;               // (the next colour starts at the next multiple of 128)
        }
        c.phase_start[ncolors] = pos;
        c.npos = pos;
        if (c.npos >= LIM - 1) { err = "too many positions"; return NSK_E_RANGE; }
        c.p_vid.assign(c.npos, -1); c.p_info.assign(c.npos, 0); c.p_slot.assign(c.npos, 0);
        c.p_cnt.assign(c.npos, 0); c.p_init.assign(c.npos, 0);
        gen_bin_start.assign((size_t)ncolors, std::vector<int64_t>(42, 0));
        c.phase_heavy_end.assign((size_t)ncolors, 0);
        for (int32_t k = 0; k < ncolors; k++) {
            gen_bin_start[k][0] = next_gen[k];
            for (int b = 0; b < 41; b++) gen_bin_start[k][b + 1] = gen_bin_start[k][b] + bin_count[k][b + 1];
            c.phase_heavy_end[k] = gen_bin_start[k][1];              // bin 0 = the hubs
        }
        for (int32_t k = 0; k < ncolors; k++) {
            for (auto &o : order[k]) {
                const int64_t p = gt_at[k]++;
                c.p_vid[p] = (int32_t)o.second;
                c.v_pos[o.second] = (int32_t)p;
                c.nsampled++;
            }
        }
        for (int64_t v = 0; v < nvar; v++) {
            const int32_t k = c.color[v];
            if (k < 0 || fast[v] == 2) continue;
            int64_t p;
            if (!fast[v]) p = gen_bin_start[k][work_bin[v]]++;
            else {
                auto it = start[k].find(sig[v]);
                if (it != start[k].end()) {
                    Cursor &cu = it->second;
                    p = cu.next++;
                    cu.rank++;
                    if (cu.pad && cu.bi < cu.pad->at.size() && cu.pad->at[cu.bi].first == cu.rank) cu.next += cu.pad->at[cu.bi++].second;
                } else {
                    auto it2 = shp[v] ? start2[k].find(shp[v]) : start2[k].end();
                    if (it2 != start2[k].end()) p = it2->second++;
                    else {
                        auto it3 = pshp[v] ? start3[k].find(pshp[v]) : start3[k].end();
                        p = (it3 != start3[k].end()) ? it3->second++ : tail_at[k]++;
                    }
                }
            }
            c.p_vid[p] = (int32_t)v;
            c.v_pos[v] = (int32_t)p;
            c.nsampled++;
        }
        return NSK_OK;
    };
    if (int rc = lay()) return rc;
    lap("positions: arrays");
    if (c.vbytes == 1 && !diag_env("NSK_NO_WIDE") && !diag_env("NSK_NO_RUN_PAD") && c.nsampled >= wide_min_variables()) {
        // the runs of the big exact classes in the layout just made
        bool any = false;
        for (const ClassAt &ca : exact_at) {
            if (ca.count < 1024) continue;
            ClassPad pd;
            if (find_run_padding(d, c, ca.start, ca.count, pd.at, pd.total)) { pads[(size_t)ca.k][ca.key] = std::move(pd); any = true; }
        }
        lap("positions: runs");
        if (any) {
            if (int rc = lay()) return rc;
            lap("positions: padded arrays");
        }
    }
    return NSK_OK;
}

// Pass 1 over the tiles: the shape of every tile.  Uniform tile = all its lanes have the same header sequence
// (slot program, draw-table candidate); shape tile = same word layout, per-lane functions and weights; general tile =
// E entries x (2 + M) words.  Phase A (parallel over tiles) classifies the tile and reduces its program to a short
// key; phase B (sequential) pools the programs, assigns stream offsets and weight rows.  Out: tile_colour[t], total4
// (the stream's size in 16-byte units).
template <typename WordsFn, typename LaneWordsFn>
static int shape_tiles(const nsk_graph_desc *d, Compiled &c, int32_t ncolors, int64_t nwb, WordsFn &&general_words,
                       LaneWordsFn &&lane_words, const std::vector<int64_t> &shape_at, const std::vector<int64_t> &shape_end,
                       int64_t shape_words, const std::vector<uint8_t> &fast, std::vector<int32_t> &tile_colour,
                       uint64_t &total4, std::string &err) {
    const int64_t nw = c.nweight, nvar = c.nvar;
    (void)nw; (void)nvar; (void)d;
    auto headers_of = [&](const std::vector<uint32_t> &w, std::vector<uint32_t> &h) {
        h.clear();
        for (size_t j = 0; j < w.size(); j += 1 + ((w[j] >> 24) & 7u)) h.push_back(w[j]);
    };
    // pass 1: shape of every tile.  Uniform tile = all its lanes have the same header sequence.
    // Phase A (parallel over tiles): classify the tile and reduce its program to a short key;
    // phase B (sequential): pool the programs, assign stream offsets and weight rows.
    struct TileShape {
        uint8_t cls;            // 0 per-lane headers, 1 general, 2 uniform, 3 shape
        uint8_t nkey;
        uint8_t empty;          // no variable at all (run padding, place_variables): takes the shape of the tile in front
        int32_t len;            // words per lane before rounding to chunks
        uint32_t flags;         // td[3]
        uint32_t nrows;         // materialised weight rows the tile needs
        uint32_t key[32];       // uniform / shape: the program words (NSK_SHAPE_WORDS); general: {E, M}
    };
    std::vector<TileShape> shapes_of((size_t)nwb);
    tile_colour.assign((size_t)nwb, 0);
    for (int32_t k = 0; k < ncolors; k++)
        for (int64_t t = c.phase_wb_base[k]; t < c.phase_wb_base[k + 1]; t++) tile_colour[t] = k;
    const bool no_shape = diag_env("NSK_NO_SHAPE") != nullptr, no_ztab = diag_env("NSK_NO_ZTAB") != nullptr;
    parallel_for(nwb, [&](int64_t tb0, int64_t tb1, int) {
        std::vector<uint32_t> words, hdrs, hdrs0;
        for (int64_t t = tb0; t < tb1; t++) {
            const int32_t k = tile_colour[t];
            const int64_t b = t - c.phase_wb_base[k];
            const int64_t p0 = c.phase_start[k] + 64 * b, p1 = std::min(p0 + 64, c.phase_fast_end[k]);
            TileShape &ts = shapes_of[t];
            memset(&ts, 0, sizeof(ts));
            bool gen_tile = false;                             // general tile (kind 6)
            for (int64_t p = p0; p < p1 && !gen_tile; p++)
                if (c.p_vid[p] >= 0 && fast[c.p_vid[p]] == 2) gen_tile = true;
            if (gen_tile) {
                // layout shared by the 64 lanes: E entries of 2 + M words, E and M the maxima over
                // the lanes
                uint32_t E = 0, M = 0, maxcard = 2;
                for (int64_t p = p0; p < p1; p++) {
                    if (c.p_vid[p] < 0) continue;
                    general_words(c.p_vid[p], &words);
                    uint32_t ne = 0;
                    for (size_t j = 0; j < words.size(); j += 2 + ((words[j + 1] >> 4) & 7u)) {
                        ne++;
                        M = std::max(M, (words[j + 1] >> 4) & 7u);
                    }
                    E = std::max(E, ne);
                    maxcard = std::max(maxcard, (uint32_t)d->variable[c.p_vid[p]].cardinality);
                }
                // the walk is specialised on M and eats whole 16-byte chunks: E is a multiple of
                // the entries per super-group (general_walk_m, nsk_kernels_gibbs.h)
                const uint32_t EG = ((2 + M) % 4 == 0) ? 1u : ((2 + M) % 2 == 0) ? 2u : 4u;
                E = (E + EG - 1) / EG * EG;
                if (c.phase_ep[k]) {            // entry-parallel group layout: no per-tile stream
                    ts.cls = 1; ts.nkey = 2; ts.key[0] = 0; ts.key[1] = M;
                    ts.len = 0;
                    ts.flags = (6u << 8) | (maxcard << 12) | (M << 16);
                    continue;
                }
                ts.cls = 1; ts.nkey = 2; ts.key[0] = E; ts.key[1] = M;
                ts.len = (int32_t)(E * (2 + M));
                ts.flags = (uint32_t)ts.len | (6u << 8) | (maxcard << 12) | (M << 16);
                if (nw * 8 > (4 << 20) && E > 0) { ts.flags |= 1u << 19; ts.nrows = E; }
                continue;
            }
            int64_t len = 0;
            bool uniform = true, have0 = false, same_shape = true;
            bool binmem = true;                // every member the lanes read is a binary variable
            // same shape: the lanes have the same number of entries and agree on which entries have members
            // at all; an entry's member slots are the most any lane has there (lanes with fewer leave null
            // words, NSK_SHAPE_NULL).  The classes of the position stage keep that padding small: exact
            // shapes first, member counts rounded up to even for the rest.
            uint32_t slots[32];
            size_t nent = 0;
            for (int64_t p = p0; p < p1; p++) {
                if (c.p_vid[p] < 0) continue;                  // padding position
                lane_words(c.p_vid[p], words);
                for (size_t j = 0; j < words.size(); j += 1 + ((words[j] >> 24) & 7u))
                    for (uint32_t m = 1; m <= ((words[j] >> 24) & 7u); m++)
                        if (d->variable[words[j + m]].cardinality != 2) binmem = false;
                len = std::max<int64_t>(len, (int64_t)words.size());
                headers_of(words, have0 ? hdrs : hdrs0);
                if (have0 && hdrs != hdrs0) {
                    uniform = false;
                    if (hdrs.size() != hdrs0.size()) same_shape = false;
                }
                {
                    const std::vector<uint32_t> &hh = have0 ? hdrs : hdrs0;
                    if (!have0) { nent = std::min<size_t>(hh.size(), 32); if (hh.size() > 32) same_shape = false; }
                    for (size_t j = 0; j < nent && j < hh.size() && same_shape; j++) {
                        const uint32_t no = (hh[j] >> 24) & 7u;
                        if (!have0) slots[j] = no;
                        else if ((no == 0) != (slots[j] == 0)) same_shape = false;
                        else slots[j] = std::max(slots[j], no);
                    }
                }
                have0 = true;
            }
            if (!have0) { hdrs0.clear(); uniform = false; same_shape = false; ts.empty = 1; }
            // (the last tile of a shape class, left with one or two lanes, is no uniform tile: a segment
            //  launch of its own per such tile costs more than the shape walk of its lanes)
            if (p0 >= shape_at[k] && p0 < shape_end[k] && same_shape) uniform = false;
            // slot program of a uniform tile: one word per member slot (an entry without other
            // members still gets one, ignored, slot):
            //   weightId | code << 24 | first << 27 | last << 28 | ignore << 29 | weight fixed << 30
            //   code: 0 NOOP, 1 IMPLY_NATURAL, 2 OR, 3 AND/ISTRUE, 4 EQUAL
            int64_t nslots = 0;
            for (uint32_t h : hdrs0) nslots += std::max<int64_t>(1, (h >> 24) & 7u);
            ts.cls = 0; ts.len = (int32_t)len;
            if (uniform && nslots <= 8 && p1 > p0) {
                uint32_t n = 0;
                for (uint32_t h : hdrs0) {
                    const int fn = (int)(h >> 27) - 1;
                    const uint32_t code = fn == 3 ? 4u : (fn == 2 || fn == 4) ? 3u : fn == 1 ? 2u : fn == 0 ? 1u : 0u;
                    const uint32_t no = (h >> 24) & 7u, wid = h & 0xFFFFFFu;
                    for (uint32_t m = 0; m < std::max(1u, no); m++)
                        ts.key[n++] = wid | (code << 24) | ((m == 0 ? 1u : 0u) << 27) |
                                      ((m + 1 >= no ? 1u : 0u) << 28) | ((no == 0 ? 1u : 0u) << 29) |
                                      ((c.w_fixed[wid] ? 1u : 0u) << 30);
                }
                ts.nkey = (uint8_t)n;
                // kind: every entry has exactly one other member and the same function code ->
                // the kernel runs a specialised, table-free step (code in bits 8..10)
                uint32_t kind = n == 0 ? 0u : (ts.key[0] >> 24) & 7u;
                for (uint32_t j = 0; j < n; j++)
                    if (((ts.key[j] >> 24) & 7u) != kind || ((ts.key[j] >> 27) & 7u) != 3u) kind = 0;   // first+last, not ignored
                // bit 11: draw-table candidate (padding slots read the always-zero id and are masked off by nslots)
                ts.cls = 2;
                ts.flags = (uint32_t)nslots | (kind << 8) | ((binmem && !no_ztab) ? 1u << 11 : 0u);
                ts.len = (int32_t)nslots;
            } else if (same_shape && len > 0 && !no_shape && [&] {
                           int64_t pl = 0;
                           for (size_t j = 0; j < nent; j++) pl += 1 + (int64_t)slots[j];
                           len = pl;                                   // (the padded length from here on)
                           return pl <= shape_words; }()) {
                // shape tile: per-lane headers (own function and weight) but one word layout for the
                // 64 lanes.  Role program, one word per stream word: 1 header | 8 header of an
                // entry without other members | 16 member | 2 first member | 4 last member; kind 7.
                uint32_t n = 0;
                for (size_t e = 0; e < nent; e++) {
                    const uint32_t no = slots[e];
                    ts.key[n++] = 1u | (no == 0 ? 8u : 0u) | 0x80000000u;   // bit 31 marks role words
                    for (uint32_t m = 0; m < no; m++)
                        ts.key[n++] = 16u | (m == 0 ? 2u : 0u) | (m + 1 == no ? 4u : 0u) | 0x80000000u;
                }
                ts.nkey = (uint8_t)n;
                ts.cls = 3;
                ts.len = (int32_t)len;
                ts.flags = (uint32_t)len | (7u << 8);
                ts.nrows = (uint32_t)hdrs0.size();
            }
        }
    }, 64);
    // a tile of padding positions only (in front of a run that starts on a quad boundary) continues the uniform tiles
    // in front of it: the segment stays one segment, its lanes sample into their own never-read positions
    for (int64_t t = 1; t < nwb; t++)
        if (shapes_of[t].empty && tile_colour[t - 1] == tile_colour[t] && shapes_of[t - 1].cls == 2) {
            shapes_of[t] = shapes_of[t - 1];
            shapes_of[t].empty = 1;
        }
    std::map<std::vector<uint32_t>, uint32_t> hdr_pool;
    std::vector<uint32_t> words, prog;
    total4 = 0;                      // stream size in 16-byte units
    const TileShape *last_ts = nullptr;
    uint32_t last_prog = 0;
    for (int64_t t = 0; t < nwb; t++) {
        const TileShape &ts = shapes_of[t];
        uint32_t *td = &c.tiles[4 * t];
        int64_t len = ts.len;
        td[2] = 0xFFFFFFFFu;
        if (ts.cls != 0) {
            if (last_ts && last_ts->cls == ts.cls && last_ts->nkey == ts.nkey &&
                !memcmp(last_ts->key, ts.key, sizeof(uint32_t) * ts.nkey)) {
                td[2] = last_prog;                              // same program as the previous tile
            } else {
                prog.clear();
                if (ts.cls == 1) {
                    // role program: 1 weight word | 32 descriptor word (8: no member slots) | 16 member
                    // slot | 2 first slot | 4 last slot
                    const uint32_t E = ts.key[0], M = ts.key[1];
                    for (uint32_t e = 0; e < E; e++) {
                        prog.push_back(1u | 0x80000000u);
                        prog.push_back(32u | (M == 0 ? 8u : 0u) | 0x80000000u);
                        for (uint32_t m = 0; m < M; m++)
                            prog.push_back(16u | (m == 0 ? 2u : 0u) | (m + 1 == M ? 4u : 0u) | 0x80000000u);
                    }
                } else {
                    prog.assign(ts.key, ts.key + ts.nkey);
                }
                auto it = hdr_pool.find(prog);
                if (it == hdr_pool.end()) {
                    it = hdr_pool.emplace(prog, (uint32_t)c.tile_hdr.size()).first;
                    c.tile_hdr.insert(c.tile_hdr.end(), prog.begin(), prog.end());
                    c.tile_hdr.resize((c.tile_hdr.size() + 7) / 8 * 8, 0u);   // pad: NOOP, weight 0
                }
                td[2] = it->second;
                last_ts = &ts; last_prog = td[2];
            }
            td[3] = ts.flags;
            if (ts.nrows) {
                // a weight table beyond the L2 (general tiles) / per-lane weights (shape tiles):
                // inference reads materialised weight rows, one coalesced row per entry
                c.tile_wrow[t] = (uint32_t)c.nwrows;
                c.nwrows += (int64_t)ts.nrows;
                if (c.nwrows >= ((int64_t)1 << 31)) { err = "weight stream too large"; return NSK_E_RANGE; }
            }
        }
        len = (len + 3) / 4 * 4;
        td[0] = (uint32_t)total4;
        td[1] = (uint32_t)len;
        total4 += (uint64_t)(len / 4) * 64;
        if (total4 >= ((uint64_t)1 << 31)) { err = "adjacency stream too large"; return NSK_E_RANGE; }
    }
    return NSK_OK;
}

int compile_graph(const nsk_graph_desc *d, Compiled &c, std::string &err) {
    const int64_t nvar = d->nvar, nfac = d->nfactor, nedge = d->nedge, nw = d->nweight;
    const int64_t nvtf = d->nvtf, nfi = d->nfactor_index;
    const int64_t LIM = (int64_t)1 << 31;
    if (nvar < 0 || nfac < 0 || nedge < 0 || nw < 0 || nvtf < 0 || nfi < 0) {
        err = "negative size in graph descriptor";
        return NSK_E_INVALID;
    }
    if (nvar >= LIM - 1 || nfac >= LIM - 1 || nedge >= LIM - 1 || nw >= LIM - 1 || nfi >= LIM - 1 ||
        nvtf >= LIM - 1) {
        err = "graph too large for 32-bit device indices";
        return NSK_E_RANGE;
    }
    if ((nvar && !d->variable) || (nfac && !d->factor) || (nedge && !d->fmap) || (nw && !d->weight) ||
        (nvtf && !d->vmap) || (nfi && !d->factor_index)) {
        err = "null array in graph descriptor";
        return NSK_E_INVALID;
    }
    c.nvar = nvar; c.nfactor = nfac; c.nedge = nedge; c.nweight = nw; c.flags = d->flags;
    // NSK_FLAG_PARTITION: the owned range is taken literally -- (0, 0) is an EMPTY shard (what the
    // reference's shard formula yields for rank 0 when nvar < ranks).  Without the flag the handle
    // samples the whole graph; a non-zero range given without the flag is honoured too (round-1 ABI).
    int64_t ob = d->own_begin, oe = d->own_end;
    if (!(d->flags & NSK_FLAG_PARTITION) && ob == 0 && oe == 0) oe = nvar;
    if (ob < 0 || oe > nvar || ob > oe) {
        err = "owned range outside [0, nvar]";
        return NSK_E_INVALID;
    }
    c.own_begin = ob; c.own_end = oe;
    const bool head_by_vid = (d->flags & NSK_FLAG_HEAD_BY_VID) != 0;
    const bool verbose = getenv("NSK_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nsk] compile %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    // ---- variables ---------------------------------------------------------------------------
    c.v_card.resize(nvar); c.v_init.resize(nvar); c.cstart.resize(nvar + 1);
    int64_t maxcard = 1, minval = 0, maxval = 0, cs = 0;
    for (int64_t v = 0; v < nvar; v++) {
        const nsk_variable &var = d->variable[v];
        if (var.cardinality < 1 || var.cardinality >= ((int64_t)1 << 22)) {
            err = fmt("variable %lld: cardinality %lld not in [1, 2^22)", v, var.cardinality);
            return NSK_E_RANGE;
        }
        int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
        if (var.vtf_offset < 0 || var.vtf_offset + nslots > nvtf) {
            err = fmt("variable %lld: vtf_offset %lld outside vmap", v, var.vtf_offset);
            return NSK_E_INDEX;
        }
        if (var.initialValue < INT32_MIN || var.initialValue > INT32_MAX) {
            err = fmt("variable %lld: initialValue does not fit int32", v);
            return NSK_E_RANGE;
        }
        // an evidence value is used as a value-slot / member index by the learning kernels
        // (learning.py:61-62 with get_factor_id_range's vmap[vtf_offset + value]): the reference
        // reads a neighbouring variable's lists or faults; here it is an error
        if (var.dataType != 0 && var.isEvidence == 1 &&
            (var.initialValue < 0 || var.initialValue >= var.cardinality)) {
            err = fmt("variable %lld: evidence value %lld outside its domain [0, %lld)", v, var.initialValue,
                      var.cardinality);
            return NSK_E_INDEX;
        }
        if (var.initialValue < 0 || var.initialValue >= var.cardinality) c.values_regular = false;
        c.v_card[v] = (int32_t)var.cardinality;
        c.v_init[v] = (int32_t)var.initialValue;
        maxcard = std::max(maxcard, var.cardinality);
        minval = std::min(minval, var.initialValue);
        maxval = std::max(maxval, var.initialValue);
        c.cstart[v] = cs;
        cs += var.cardinality == 2 ? 1 : var.cardinality;      // factorgraph.py:41-45
    }
    c.cstart[nvar] = cs;
    if (cs >= LIM - 1) {
        err = "tally array too large for 32-bit device indices";
        return NSK_E_RANGE;
    }
    c.ncount = cs;
    c.vbytes = (maxcard <= 127 && minval >= -128 && maxval <= 127) ? 1 : 4;

    // ---- weights -----------------------------------------------------------------------------
    c.w_init.resize(nw); c.w_fixed.resize(nw);
    for (int64_t i = 0; i < nw; i++) {
        c.w_init[i] = d->weight[i].initialValue;
        c.w_fixed[i] = d->weight[i].isFixed ? 1 : 0;
    }

    // ---- factors and edges: narrow copies; validated lazily for factors that are reachable ----
    c.f_rec.assign((size_t)nfac * 4 + 4, 0); c.f_feat.resize(nfac);
    for (int64_t f = 0; f < nfac; f++)
        if (d->factor[f].arity >= ((int64_t)1 << 24)) {
            err = fmt("factor %lld: arity %lld too large", f, d->factor[f].arity);
            return NSK_E_RANGE;
        }
    parallel_for(nfac, [&](int64_t fb0, int64_t fb1, int) {
    for (int64_t f = fb0; f < fb1; f++) {
        const nsk_factor &fa = d->factor[f];
        int64_t ar = fa.arity;
        if (ar < 0) ar = 0;
        c.f_rec[4 * f] = ((uint32_t)ar << 8) | (uint32_t)((fa.factorFunction + 1) & 0xff);
        c.f_rec[4 * f + 1] = (uint32_t)(int32_t)std::max<int64_t>(std::min<int64_t>(fa.ftv_offset, LIM - 2), -1);
        c.f_rec[4 * f + 2] = (uint32_t)(int32_t)std::max<int64_t>(std::min<int64_t>(fa.weightId, LIM - 2), -1);
        c.f_feat[f] = fa.featureValue;
    }
    });
    c.m_rec.assign((size_t)nedge * 2 + 2, 0);
    parallel_for(nedge, [&](int64_t lb0, int64_t lb1, int) {
    for (int64_t l = lb0; l < lb1; l++) {
        int64_t vid = d->fmap[l].vid, deo = d->fmap[l].dense_equal_to;
        c.m_rec[2 * l] = (vid < 0 || vid >= nvar) ? -1 : (int32_t)vid;
        c.m_rec[2 * l + 1] = (int32_t)std::max<int64_t>(std::min<int64_t>(deo, INT32_MAX), INT32_MIN);
    }
    });

    lap("records");
    // ---- which variables does this handle sample? -------------------------------------------
    std::vector<uint8_t> sampled(nvar, 0);
    for (int64_t v = ob; v < oe; v++) sampled[v] = d->variable[v].isEvidence != 4;   // inference.py:21-23

    int64_t max_ratio_arity = 0;
    if (int vrc = validate_reachable(d, c, sampled, head_by_vid, max_ratio_arity, err)) return vrc;
    lap("validate");
    c.logtab.resize((size_t)max_ratio_arity + 2);
    c.logtab[0] = 0.0;
    for (size_t k = 1; k < c.logtab.size(); k++) c.logtab[k] = std::log((double)k);   // math.log, inference.py:222

    // ---- colouring: no two variables of a colour may read each other ---------------------------
    // reads(v) = members of every factor in v's lists (+ the literal head index variable)
    auto for_each_read_slow = [&](int64_t v, auto &&fn_) {
        const nsk_variable &var = d->variable[v];
        const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
        for (int64_t k = 0; k < nslots; k++) {
            const nsk_vtf &vt = d->vmap[var.vtf_offset + k];
            for (int64_t j = 0; j < vt.factor_index_length; j++) {
                const int64_t f = d->factor_index[vt.factor_index_offset + j];
                const nsk_factor &fa = d->factor[f];
                const int fnid = fa.factorFunction;
                if (fnid == -1) continue;
                int64_t need = (fnid == 21 || fnid == 22 || fnid == 25 || fnid == 26) ? 2
                             : (fnid == 23 || fnid == 24) ? 3 : (fnid >= 18 && fnid <= 20) ? 1 : 0;
                int64_t s = fa.ftv_offset, e = std::max(s + fa.arity, s + need);
                if (fnid == 30) e = std::max(e, s + d->variable[d->fmap[s].vid].cardinality - 1);
                for (int64_t l = s; l < e; l++) fn_(d->fmap[l].vid);
                if (literal_head_function(fnid) && !head_by_vid) fn_(s + fa.arity - 1);
            }
        }
    };

    // compact read lists: reads of v, sorted and unique, self excluded, as int32 -- built once, in
    // parallel, from the packed records; every colouring pass below walks these 4-byte lists
    // instead of chasing vmap -> factor_index -> factor -> fmap again
    std::vector<int64_t> rd_off((size_t)nvar + 1, 0);
    std::vector<int32_t> rd_len((size_t)nvar, 0);
    std::vector<int32_t> rd;
    bool use_rd = true;
    {
        parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
            for (int64_t v = b0; v < b1; v++) {
                if (!sampled[v]) continue;
                int64_t n = 0;
                for_each_read_slow(v, [&](int64_t b) { if (b != v) n++; });
                rd_off[v + 1] = n;
            }
        });
        for (int64_t v = 0; v < nvar; v++) rd_off[v + 1] += rd_off[v];
        // (factors with a huge arity make the lists quadratic: beyond 2^32 entries walk the records)
        use_rd = rd_off[nvar] < ((int64_t)1 << 32);
        rd.resize(use_rd ? (size_t)rd_off[nvar] : 0);
        if (use_rd) parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
            for (int64_t v = b0; v < b1; v++) {
                if (!sampled[v]) continue;
                int32_t *out = rd.data() + rd_off[v];
                int64_t n = 0;
                for_each_read_slow(v, [&](int64_t b) { if (b != v) out[n++] = (int32_t)b; });
                std::sort(out, out + n);
                rd_len[v] = (int32_t)(std::unique(out, out + n) - out);
            }
        });
    }
    auto for_each_read = [&](int64_t v, auto &&fn_) {
        if (!use_rd) { for_each_read_slow(v, fn_); return; }
        const int32_t *p = rd.data() + rd_off[v];
        for (int32_t j = 0, n = rd_len[v]; j < n; j++) fn_((int64_t)p[j]);
    };
    lap("read lists");
    int32_t ncolors = colour_sampled(c, sampled, for_each_read, lap);
    lap("balancing");
    // ---- ghosts: variables outside the owned range read by a sampled variable -------------------
    if (ob > 0 || oe < nvar) {
        std::vector<uint8_t> need(nvar, 0);
        for (int64_t v = 0; v < nvar; v++)
            if (sampled[v]) for_each_read(v, [&](int64_t b) { if (b < ob || b >= oe) need[b] = 1; });
        for (int64_t v = 0; v < nvar; v++) if (need[v]) c.ghost_needs.push_back((int32_t)v);
    }

    lap("ghosts");
    // ---- fast-path eligibility (DESIGN.md "fast path"): a binary dataType-0 variable whose every
    // factor is a symmetric boolean function it is a member of, with <= 6 other members and a
    // weight id below 2^24; and featureValue == 1 so that learning can use the same stream.
    std::vector<uint8_t> fast(nvar, 0);
    auto fast_function = [](int fn) { return fn == -1 || (fn >= 0 && fn <= 4); };
    const bool no_fast = diag_env("NSK_NO_FAST") != nullptr;      // diagnostic: everything on the generic path
    parallel_for(nvar, [&](int64_t vb0, int64_t vb1, int) {
    for (int64_t v = vb0; v < vb1; v++) {
        if (c.color[v] < 0 || no_fast) continue;
        const nsk_variable &var = d->variable[v];
        if (var.cardinality != 2 || var.dataType != 0) continue;
        const nsk_vtf &vt = d->vmap[var.vtf_offset];
        bool ok = vt.factor_index_length <= 4096;
        for (int64_t j = 0; ok && j < vt.factor_index_length; j++) {
            const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
            if (!fast_function(fa.factorFunction) || fa.weightId >= (1 << 24) || fa.featureValue != 1.0 ||
                fa.arity > 64) { ok = false; break; }
            if (fa.factorFunction == -1) continue;
            int64_t others = 0;
            bool member = false;
            for (int64_t l = fa.ftv_offset; l < fa.ftv_offset + fa.arity; l++) {
                if (d->fmap[l].vid == v) member = true; else others++;
            }
            if (!member || others > 6) ok = false;
        }
        fast[v] = ok;
    }
    });

    lap("fast eligibility");
    // ---- general tiles (kind 6): variables of cardinality <= 8 and any dataType whose factors are
    // boolean symmetric functions, IMPLY_MLN or the categorical *_CAT functions.  Stream words per
    // entry: W0 = weight id; W1 = code | others << 4 | own role << 7 (1 body, 2 head of a positional
    // function) | own dense_equal_to << 9 | owning candidate value << 14 (15 = every candidate,
    // dataType 0; 14 = none, padding entry); then one word per other member: id |
    // dense_equal_to << 27 (id NSK_GEN_NULL = empty slot).  Every such entry evaluates to
    // (candidate == c) ? A : B with c, A, B known once the other members have been read.
    auto general_code = [](int fn) -> int {
        switch (fn) {
        case -1: return 0; case 0: return 1; case 1: return 2; case 2: case 4: return 3; case 3: return 4;
        case 13: return 5; case 12: case 15: return 6; case 14: return 7; case 16: return 8; case 17: return 9;
        default: return -1;
        }
    };
    // (general-tile member words keep the id in 27 bits: positions include padding, so stay well below)
    const bool no_general = diag_env("NSK_NO_GENERAL") != nullptr || nvar >= (int64_t)100000000;
    // longer lists go to the wave-per-variable kernel: a tile is walked by one wave, so its longest
    // lane sets a serial chain of memory round trips and the longest tile the kernel's run time
    const int64_t gen_max_entries = diag_env("NSK_GEN_MAX_ENTRIES") ? std::max(1, std::min(24, atoi(diag_env("NSK_GEN_MAX_ENTRIES")))) : 16;
    // (hub = true lifts the per-lane size caps: the entry-parallel hub kernels take up to 256 entries)
    //
    // The variable may occur SEVERAL times in one factor (the config-#5 generator draws the other
    // members from [v - 1024, v + 1024], v included): as body member and head of a positional
    // function, or with different dense_equal_to values.  With x = the candidate value c at every
    // own edge, eval_factor still reduces to (c == cstar) ? A : B:
    //   * positional function, own body edges (all with dense_equal_to db) AND own head (dh) -> role 3:
    //     the head test is the constant (db == dh) [IMPLY_MLN: true -- the head is only reached with
    //     every body member, the variable included, non-zero], the body test is over the other members;
    //   * own edges whose dense_equal_to disagree: the variable cannot match all of them -- AND_CAT /
    //     EQUAL_CAT_CONST and IMPLY_NATURAL_CAT (own body edges) are constant 0, IMPLY_MLN_CAT (own
    //     body edges) constant 1, OR_CAT over a binary variable with both values named constant 1
    //     (codes 10 / 11; no member words); OR_CAT naming two of more than two values is not of the
    //     one-cstar form and keeps the variable on the generic path.
    // A dataType-1 variable finds such a factor in the list of EVERY dense_equal_to its own edges
    // name (dataloading.py:34-38); the learning sweep visits a factor once per variable
    // (learning.py:76-95), so the entry in the list of the larger value names the smaller one as its
    // `partner` (descriptor bits 19-22) and is skipped when the partner's list is selected too.
    // the weight's slot in the device table (nsk_compile.h wmap; the caller's id until the numbering exists:
    // eligibility and the shapes of pass 1 never look at a direct weight's id)
    auto slot_of_weight = [&](int64_t wid) -> uint32_t {
        return (c.wmap.empty() || wid < 0 || wid >= nw) ? (uint32_t)wid : (uint32_t)c.wmap[(size_t)wid];
    };
    auto general_words_walk = [&](int64_t v, std::vector<uint32_t> *out, bool hub, size_t hub_cap) -> bool {
        const nsk_variable &var = d->variable[v];
        if (var.cardinality > 8 || var.cardinality < 2) return false;
        // (an evidence value outside the domain is kept off the tiles: their saved facts hold the
        // variable's own values in 4 bits)
        if (var.initialValue < 0 || var.initialValue >= var.cardinality) return false;
        const int64_t nslots = var.dataType == 0 ? 1 : var.cardinality;
        size_t nwords = 0, nentries = 0;
        if (out) out->clear();
        for (int64_t k = 0; k < nslots; k++) {
            const nsk_vtf &vt = d->vmap[var.vtf_offset + k];
            for (int64_t j = 0; j < vt.factor_index_length; j++) {
                const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
                int code = general_code(fa.factorFunction);
                if (code < 0 || fa.featureValue != 1.0 || fa.arity > 64) return false;
                const bool positional = code == 5 || code == 8 || code == 9;
                const bool cat = code >= 6;
                const int64_t s = fa.ftv_offset, e = s + fa.arity;
                int64_t others = 0, self_body = 0, self_head = 0;
                int64_t body_deo = -1, head_deo = -1;      // dense_equal_to of the own body edges / own head
                bool body_deo_mixed = false;
                int64_t own_deo[2] = {-1, -1};             // distinct dense_equal_to of all own edges
                int n_own_deo = 0;
                uint32_t mem[8];
                const bool keyed = cat || var.dataType != 0;      // own dense_equal_to matters
                for (int64_t l = s; l < e; l++) {
                    const int64_t vid = d->fmap[l].vid, deo = d->fmap[l].dense_equal_to;
                    if (vid == v) {
                        if (keyed) {
                            if (cat && (deo < 0 || deo > 31)) return false;
                            if (n_own_deo == 0 || (own_deo[0] != deo && (n_own_deo < 2 || own_deo[1] != deo))) {
                                if (n_own_deo == 2) return false;          // three different own values: generic path
                                own_deo[n_own_deo++] = deo;
                            }
                        }
                        if (code == 0) continue;
                        if (positional && l == e - 1) { self_head++; head_deo = deo; }
                        else {
                            self_body++;
                            if (body_deo >= 0 && body_deo != deo) body_deo_mixed = true;
                            body_deo = deo;
                        }
                    } else if (code != 0) {
                        if (others >= 6) return false;
                        int64_t rd = vid;                                  // index the value is read at
                        if (positional && l == e - 1 && !head_by_vid) rd = l;   // inference.py:243,277,292
                        if (rd >= (int64_t)NSK_GEN_NULL) return false;
                        int64_t dd = cat ? deo : 0;
                        if (dd < 0 || dd > 31) return false;
                        mem[others++] = (uint32_t)rd | ((uint32_t)dd << 27);
                    }
                }
                if (code != 0 && self_body + self_head == 0) return false;
                // a non-categorical function over a dataType-1 variable whose own edges disagree:
                // rare and not of the tile form (the lists are keyed by values the function ignores)
                if (!cat && var.dataType != 0 && n_own_deo > 1 && code != 0) return false;
                if (code == 0 && var.dataType != 0 && n_own_deo > 1) return false;
                uint32_t role = 0, hbit = 0;
                int64_t self_deo = keyed && n_own_deo > 0 ? own_deo[0] : -1;
                if (code != 0 && positional) {
                    if (self_body > 0 && cat && body_deo_mixed) {            // the body can never match
                        code = code == 9 ? 11 : 10;
                        others = 0;
                    } else if (self_body > 0 && self_head > 0) {
                        role = 3; self_deo = body_deo;
                        hbit = cat ? (body_deo == head_deo ? 1u : 0u) : 1u;
                    } else if (self_body > 0) { role = 1; self_deo = body_deo; }
                    else { role = 2; self_deo = head_deo; }
                } else if (code != 0 && cat && n_own_deo > 1) {             // AND_CAT / EQUAL_CAT_CONST / OR_CAT
                    if (code == 6) { code = 10; others = 0; }
                    else if (var.cardinality == 2) { code = 11; others = 0; }     // own edges name 0 and 1
                    else return false;
                }
                uint32_t partner = 0;                       // bit 19: has one; bits 20-22: its value
                if (var.dataType != 0 && n_own_deo > 1) {
                    const int64_t lo = std::min(own_deo[0], own_deo[1]), hi = std::max(own_deo[0], own_deo[1]);
                    if (lo < 0 || hi > 7) return false;
                    if (k == hi) partner = 1u | ((uint32_t)lo << 1);
                }
                const uint32_t kslot = var.dataType == 0 ? 15u : (uint32_t)k;
                nwords += 2 + (size_t)others;
                if (hub ? (++nentries > (hub_cap ? hub_cap : 256)) : (nwords > 120 || (int64_t)++nentries > gen_max_entries)) return false;
                if (out) {
                    out->push_back((uint32_t)fa.weightId);          // (the caller's id: general_words numbers it)
                    out->push_back((uint32_t)code | ((uint32_t)others << 4) | (role << 7) |
                                   ((uint32_t)(cat && self_deo > 0 ? self_deo : 0) << 9) | (kslot << 14) |
                                   (hbit << 18) | (partner << 19));
                    for (int64_t m = 0; m < others; m++) out->push_back(mem[m]);
                }
            }
        }
        return true;
    };
    // The entry lists are read five times on the way to the streams (eligibility, lane order, tile shapes, the two
    // passes of the entry-parallel groups), the later ones in position order, where a walk through the caller's
    // records -- variable, value slots, factor ids, factors, members: six to ten cache lines a variable -- has no
    // locality left (50M LR graph: 4.7 - 5.6 s a pass against 1.1 s in id order).  The eligibility pass keeps what it
    // found: the words of every variable it sends to the general tiles, id order, one or two cache lines a variable.
    // One chunk per thread of that pass, read where it was written (a flat copy would fault the pages in twice).
    std::vector<std::vector<uint32_t>> gw_chunk;           // the words, id order inside a chunk
    std::vector<int64_t> gw_v0;                            // first variable of every chunk (ascending)
    std::vector<uint32_t> gw_at;                           // [nvar] start inside the variable's chunk
    std::vector<uint8_t> gw_len;                           // [nvar] words (a lane's list is at most 120); 0: not kept
    auto general_words = [&](int64_t v, std::vector<uint32_t> *out, bool hub = false, size_t hub_cap = 0) -> bool {
        if (out && !hub && !gw_len.empty() && gw_len[(size_t)v]) {
            const size_t t = (size_t)(std::upper_bound(gw_v0.begin(), gw_v0.end(), v) - gw_v0.begin()) - 1;
            const uint32_t *src = gw_chunk[t].data() + gw_at[(size_t)v];
            out->assign(src, src + gw_len[(size_t)v]);
        } else if (!general_words_walk(v, out, hub, hub_cap)) return false;
        if (out && !c.wmap.empty())             // the weight's slot in the device table, once the numbering exists
            for (size_t j = 0; j < out->size(); j += 2 + (((*out)[j + 1] >> 4) & 7u)) (*out)[j] = slot_of_weight((int64_t)(*out)[j]);
        return true;
    };
    {
        const bool keep = !diag_env("NSK_NO_WORD_CACHE") && !no_fast && !no_general;
        const size_t T = (size_t)compile_threads();
        std::vector<uint8_t> overflow(T, 0);
        if (keep) { gw_chunk.resize(T); gw_v0.assign(T, nvar); gw_at.resize((size_t)nvar); gw_len.assign((size_t)nvar, 0); }
        parallel_for(nvar, [&](int64_t vb0, int64_t vb1, int t) {
            std::vector<uint32_t> w;
            if (keep) { gw_v0[(size_t)t] = vb0; gw_chunk[(size_t)t].reserve((size_t)(vb1 - vb0) * 12); }
            for (int64_t v = vb0; v < vb1; v++) {
                if (c.color[v] < 0 || fast[v] || no_fast || no_general || !general_words_walk(v, keep ? &w : nullptr, false, 0)) continue;
                fast[v] = 2;
                if (!keep || overflow[(size_t)t]) continue;
                std::vector<uint32_t> &ch = gw_chunk[(size_t)t];
                if (ch.size() + w.size() > (size_t)0xFFFFFFFFu) { overflow[(size_t)t] = 1; continue; }
                gw_at[(size_t)v] = (uint32_t)ch.size();
                gw_len[(size_t)v] = (uint8_t)w.size();
                ch.insert(ch.end(), w.begin(), w.end());
            }
        });
        // (parallel_for hands out ascending ranges: gw_v0 is ascending, threads that took no part keep nvar at its end)
    }

    if (verbose) { int64_t n2 = 0, n1 = 0; for (int64_t v = 0; v < nvar; v++) { n2 += fast[v] == 2; n1 += fast[v] == 1; } fprintf(stderr, "[nsk] eligibility: %lld general-tile variables (entry lists kept), %lld fast\n", (long long)n2, (long long)n1); }
    lap("general eligibility");
    // ---- positions: colour-major.  Inside a colour: the fast variables grouped by "shape class"
    // -- the sequence of (function, member count, weight id) of their factor lists plus their
    // evidence flag -- so that the 64 lanes of a tile share one slot program; every class with at
    // least 64 members starts on a tile boundary (the gap is padded with empty positions,
    // p_vid = -1); smaller classes share a tail in id order; then the generic-path variables.
    // Order inside a class: variable id.
    c.phase_start.assign((size_t)ncolors + 1, 0);
    c.phase_end.assign((size_t)ncolors, 0);
    c.phase_fast_end.assign((size_t)ncolors, 0);
    c.phase_gen_tile.assign((size_t)ncolors, 0);
    c.v_pos.assign(nvar, -1);
    const int64_t shape_words = diag_env("NSK_SHAPE_MAX_WORDS") ? std::max<int64_t>(4, std::min<int64_t>(32, atoll(diag_env("NSK_SHAPE_MAX_WORDS")))) : NSK_SHAPE_WORDS;
    const int64_t shape_words_ep = diag_env("NSK_SHAPE_MAX_WORDS") ? shape_words : NSK_SHAPE_WORDS_EP;
    // [shape_at[k], shape_end[k]): the positions of colour k's shape classes (tile shapes, pass 1)
    std::vector<int64_t> shape_at((size_t)ncolors, 0), shape_end((size_t)ncolors, 0);
        if (int prc = place_variables(d, c, ncolors, fast, general_words, no_general, shape_words, shape_words_ep, shape_at, shape_end, lap, err))
        return prc;
    // ---- internal ids: a positioned variable's id is its position; the others (ghosts, isEvidence
    // == 4 -- read but never sampled here) follow.  Every variable id stored for the device from
    // here on is internal (m_rec, tiles, gstream, v_card): values are kept in this order, so the
    // stores of a colour class are contiguous and its gathers run through the other classes' ranges
    // in step with the lanes (DESIGN.md "internal numbering").
    c.iid.assign(nvar, -1);
    {
        // (the ghosts the sampled variables READ come first among the others, in id order: the receive list of
        // a peer-to-peer exchange -- all of them, ascending -- is then one contiguous run of internal ids, which
        // lets a shard's kernels read ghost values straight from the exchange buffer, nsk_api.hip)
        int64_t next = c.npos;
        for (int32_t v : c.ghost_needs) if (c.v_pos[v] < 0) c.iid[v] = (int32_t)next++;
        for (int64_t v = 0; v < nvar; v++)
            if (c.v_pos[v] >= 0) c.iid[v] = c.v_pos[v];
            else if (c.iid[v] < 0) c.iid[v] = (int32_t)next++;
        // one more id that belongs to no variable and always holds 0: where the ignored member slots and the
        // padding of uniform tiles point.  The draw-table kernels take a member's value as its bit (values
        // are regular, members binary), so such a slot must not read a categorical variable's value --
        // position 0 may hold one (a 2 there set the NEXT slot's bit: wrong table entry, wrong gradient).
        c.zero_id = next++;
        c.nid = next;
        if (c.nid >= LIM - 1) { err = "too many internal ids"; return NSK_E_RANGE; }
        parallel_for(nedge, [&](int64_t lb0, int64_t lb1, int) {
            for (int64_t l = lb0; l < lb1; l++)
                if (c.m_rec[2 * l] >= 0) c.m_rec[2 * l] = c.iid[c.m_rec[2 * l]];
        });
        std::vector<int32_t> card_i((size_t)c.nid, 2);
        for (int64_t v = 0; v < nvar; v++) card_i[c.iid[v]] = c.v_card[v];
        c.v_card_i.swap(card_i);
    }
    lap("positions");
    // ---- inlined adjacency streams of the fast variables, one column-major tile per 64 positions
    c.phase_wb_base.assign((size_t)ncolors + 1, 0);
    for (int32_t k = 0; k < ncolors; k++)
        c.phase_wb_base[k + 1] = c.phase_wb_base[k] + (c.phase_fast_end[k] - c.phase_start[k] + 63) / 64;
    const int64_t nwb = c.phase_wb_base[ncolors];
    c.tiles.assign((size_t)nwb * 4 + 4, 0);
    c.tile_wrow.assign((size_t)nwb + 1, 0);
    {
        // words of one lane: per factor of the variable, in list order, a header then the ids of
        // the members other than the variable itself
        auto lane_words = [&](int64_t v, std::vector<uint32_t> &out) {
            out.clear();
            const nsk_variable &var = d->variable[v];
            const nsk_vtf &vt = d->vmap[var.vtf_offset];
            for (int64_t j = 0; j < vt.factor_index_length; j++) {
                const nsk_factor &fa = d->factor[d->factor_index[vt.factor_index_offset + j]];
                const size_t at = out.size();
                out.push_back(0);
                uint32_t others = 0;
                if (fa.factorFunction != -1)
                    for (int64_t l = fa.ftv_offset; l < fa.ftv_offset + fa.arity; l++)
                        if (d->fmap[l].vid != v) { out.push_back((uint32_t)d->fmap[l].vid); others++; }
                out[at] = ((uint32_t)(fa.factorFunction + 1) << 27) | (others << 24) | slot_of_weight(fa.weightId);
            }
        };
        std::vector<int32_t> tile_colour;
        uint64_t total4 = 0;
        if (int trc = shape_tiles(d, c, ncolors, nwb, general_words, lane_words, shape_at, shape_end, shape_words, fast, tile_colour, total4, err))
            return trc;
        lap("tile shapes (pass 1)");
        c.tile_hdr.resize(c.tile_hdr.size() + 8, 0u);
        find_direct_weights(d, c, nwb, verbose);
        if (number_direct_weights(d, c)) lap("weight numbering");
        if (int src = plan_segments(d, c, ncolors, lane_words, fast, verbose, err)) return src;
        lap("segments");
        fill_tiles(d, c, nwb, total4, tile_colour, general_words, lane_words);
    }
    lap("tile fill (pass 2)");
    if (int erc = build_ep_groups(d, c, ncolors, general_words, lap, verbose, err)) return erc;
    lap("entry-parallel groups");
    if (int arc = build_segment_adjacency(c, err)) return arc;
    if (int wrc = build_segment_wide(c, err)) return wrc;
    if (int hrc = build_hub_streams(d, c, ncolors, general_words, no_general, verbose, err)) return hrc;
    lap("compact streams");
    plan_learning_launches(c, ncolors);
    return build_index_and_census(d, c, ncolors, head_by_vid, for_each_read, lap, err);
}

}  // namespace nsk
