// nsk_host.cpp -- host-only entry points: the inverted-index build and the graph.factors parser.
//
//   nsk_compute_var_map  <-  numbskull/dataloading.py:16-81  compute_var_map
//   nsk_parse_factors    <-  numbskull/dataloading.py:196-235 load_factors
//   nsk_parse_domains    <-  numbskull/dataloading.py:159-187 load_domains
//   nsk_write_probabilities <- numbskull/factorgraph.py:216-229 dump_probabilities
//
// These run once per graph load (not per sweep) and need no GPU.  They reproduce the reference's
// output arrays exactly, including the quirks a caller can observe: offsets are not compacted
// after de-duplication, slot lengths count the edges of skipped factors, and slices are clipped
// to the factor_index array like numpy slices are.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"
#include "nsk_compile.h"          // nsk::parallel_for (host threads)

namespace nsk { void set_error(const std::string &m); }   // nsk_api.hip (thread-local message)

static inline uint64_t be64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}
static inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

extern "C" int nsk_compute_var_map(int64_t nvar, const nsk_variable *variable, int64_t nfactor,
                                   const nsk_factor *factor, int64_t nedge, const nsk_ftv *fmap,
                                   int64_t nvtf, nsk_vtf *vmap, int64_t nfi, int64_t *factor_index,
                                   const uint8_t *domain_mask, const int64_t *factors_to_skip,
                                   int64_t nskip) {
    auto slot_of = [&](int64_t l, int64_t &idx) -> bool {
        const int64_t vid = fmap[l].vid;
        if (vid < 0 || vid >= nvar) return false;
        const int64_t val = variable[vid].dataType == 1 ? fmap[l].dense_equal_to : 0;
        idx = variable[vid].vtf_offset + val;
        return idx >= 0 && idx < nvtf;
    };
    // implicit domains (dataloading.py:21-30)
    for (int64_t i = 0; i < nvar; i++) {
        if (variable[i].dataType == 0 || (domain_mask && domain_mask[i])) continue;
        if (variable[i].vtf_offset < 0 || variable[i].vtf_offset + variable[i].cardinality > nvtf) {
            nsk::set_error("compute_var_map: vtf_offset outside vmap");
            return NSK_E_INDEX;
        }
        for (int64_t k = 0; k < variable[i].cardinality; k++) vmap[variable[i].vtf_offset + k].value = k;
    }
    // lengths over EVERY edge (34-38), then exclusive prefix (41-46).  The edges go over the host threads (atomic
    // increments: a slot's edges are spread over the factors): the counts do not depend on the order
    {
        std::vector<int64_t> added((size_t)nvtf, 0);          // (the caller's records are packed: count beside them)
        std::vector<uint8_t> bad((size_t)nsk::compile_threads(), 0);
        nsk::parallel_for(nedge, [&](int64_t l0, int64_t l1, int t) {
            for (int64_t l = l0; l < l1; l++) {
                int64_t idx;
                if (!slot_of(l, idx)) { bad[(size_t)t] = 1; return; }
                __atomic_fetch_add(&added[(size_t)idx], (int64_t)1, __ATOMIC_RELAXED);
            }
        }, 1 << 16);
        for (uint8_t x : bad)
            if (x) { nsk::set_error("compute_var_map: edge refers outside variables/vmap"); return NSK_E_INDEX; }
        nsk::parallel_for(nvtf, [&](int64_t i0, int64_t i1, int) {
            for (int64_t i = i0; i < i1; i++) vmap[i].factor_index_length += added[(size_t)i];
        }, 1 << 16);
    }
    int64_t last_len = 0, last_off = 0;
    for (int64_t i = 0; i < nvtf; i++) {
        vmap[i].factor_index_offset = last_off + last_len;
        last_len = vmap[i].factor_index_length;
        last_off = vmap[i].factor_index_offset;
    }
    // scatter the factor ids, skipping factors_to_skip (49-65).  The reference walks the factors in order; every
    // slot is sorted below, so only WHICH ids land in a slot matters, not the order they arrive in: the factors go
    // over the host threads with an atomic cursor per slot.  (The skip list is consumed by the reference's own
    // one-pointer walk -- an id out of order is never skipped -- into a flag per factor first.)
    std::vector<int64_t> cursor((size_t)nvtf);
    nsk::parallel_for(nvtf, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; i++) cursor[(size_t)i] = vmap[i].factor_index_offset;
    }, 1 << 16);
    std::vector<uint8_t> skipped;
    if (nskip > 0) {
        skipped.assign((size_t)nfactor, 0);
        int64_t fts = 0;
        for (int64_t f = 0; f < nfactor; f++)
            if (fts < nskip && factors_to_skip[fts] == f) { skipped[(size_t)f] = 1; fts++; }
    }
    {
        std::vector<uint8_t> bad((size_t)nsk::compile_threads(), 0);
        nsk::parallel_for(nfactor, [&](int64_t f0, int64_t f1, int t) {
            for (int64_t f = f0; f < f1; f++) {
                if (!skipped.empty() && skipped[(size_t)f]) continue;
                const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
                if (s < 0 || e > nedge) { bad[(size_t)t] = 1; return; }
                for (int64_t l = s; l < e; l++) {
                    int64_t idx;
                    slot_of(l, idx);
                    const int64_t at = __atomic_fetch_add(&cursor[(size_t)idx], (int64_t)1, __ATOMIC_RELAXED);
                    if (at >= nfi) { bad[(size_t)t] = 2; return; }
                    factor_index[at] = f;
                }
            }
        }, 1 << 14);
        for (uint8_t x : bad) {
            if (x == 1) { nsk::set_error("compute_var_map: factor members outside fmap"); return NSK_E_INDEX; }
            if (x == 2) { nsk::set_error("compute_var_map: index out of bounds for factor_index (IndexError in the reference)"); return NSK_E_INDEX; }
        }
    }
    // per slot: sort, drop duplicates in place, shrink the length (68-81); the slots' ranges are disjoint
    nsk::parallel_for(nvtf, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; i++) {
            int64_t off = vmap[i].factor_index_offset, len = vmap[i].factor_index_length;
            if (off > nfi) off = nfi;
            if (off + len > nfi) len = nfi - off;
            int64_t *lst = factor_index + off;
            if (!std::is_sorted(lst, lst + len)) std::sort(lst, lst + len);
            int64_t n = 0, last = -1;
            for (int64_t k = 0; k < len; k++) {
                if (lst[k] == last) continue;
                last = lst[k];
                lst[n++] = last;
            }
            vmap[i].factor_index_length = n;
        }
    }, 1 << 14);
    return NSK_OK;
}

// FactorGraph.__init__'s arrays (factorgraph.py:41-53): cstart = cumulative tally slots (one for a binary variable,
// `cardinality` otherwise), the variables' initial values as a dense vector (what var_value / var_value_evid are tiled
// from), the largest cardinality (Z's width) and the longest factor list (fids' width).  The caller's records are
// packed 27-byte structs: numpy's strided field reads take seconds at 50M variables; here the host threads walk them.
extern "C" int nsk_state_layout(int64_t nvar, const nsk_variable *variable, int64_t nvtf, const nsk_vtf *vmap,
                                int64_t *cstart, int64_t *init, int64_t *max_card, int64_t *longest) {
    if (nvar < 0 || nvtf < 0 || (nvar && (!variable || !cstart || !init))) { nsk::set_error("nsk_state_layout: null argument"); return NSK_E_INVALID; }
    const size_t T = (size_t)nsk::compile_threads();
    std::vector<int64_t> part(T + 1, 0), mc(T, 0), ml(T, 0);
    cstart[0] = 0;
    auto slots = [&](int64_t v) { const int64_t c = variable[v].cardinality; return c == 2 ? (int64_t)1 : c; };
    nsk::parallel_for(nvar, [&](int64_t v0, int64_t v1, int t) {
        int64_t sum = 0, m = 0;
        for (int64_t v = v0; v < v1; v++) {
            sum += slots(v);
            cstart[v + 1] = sum;                          // (local prefix; the block's base is added below)
            init[v] = variable[v].initialValue;
            m = std::max<int64_t>(m, variable[v].cardinality);
        }
        part[(size_t)t + 1] = sum; mc[(size_t)t] = m;
    }, 1 << 16);
    for (size_t t = 0; t < T; t++) part[t + 1] += part[t];
    nsk::parallel_for(nvar, [&](int64_t v0, int64_t v1, int t) {
        const int64_t base = part[(size_t)t];
        if (base) for (int64_t v = v0; v < v1; v++) cstart[v + 1] += base;
    }, 1 << 16);
    nsk::parallel_for(nvtf, [&](int64_t i0, int64_t i1, int t) {
        int64_t m = 0;
        for (int64_t i = i0; i < i1; i++) m = std::max<int64_t>(m, vmap[i].factor_index_length);
        ml[(size_t)t] = m;
    }, 1 << 16);
    if (max_card) { *max_card = 0; for (int64_t m : mc) *max_card = std::max(*max_card, m); }
    if (longest) { *longest = 0; for (int64_t m : ml) *longest = std::max(*longest, m); }
    return NSK_OK;
}

extern "C" int nsk_parse_factors(const uint8_t *data, int64_t nbytes, int64_t nfactor, int64_t nedge,
                                 nsk_factor *factor, nsk_ftv *fmap, const uint8_t *domain_mask,
                                 const nsk_variable *variable, int64_t nvar, const nsk_vtf *vmap) {
    int64_t index = 0, e = 0;
    for (int64_t i = 0; i < nfactor; i++) {
        if (index + 10 > nbytes) { nsk::set_error("graph.factors truncated"); return NSK_E_INDEX; }
        factor[i].factorFunction = (int16_t)be16(data + index);
        const int64_t arity = (int64_t)be64(data + index + 2);
        factor[i].arity = arity;
        factor[i].ftv_offset = e;
        index += 10;
        if (arity < 0 || e + arity > nedge || index + 16 * arity + 16 > nbytes) {
            nsk::set_error("graph.factors: record runs past the file or the edge count in graph.meta");
            return NSK_E_INDEX;
        }
        for (int64_t k = 0; k < arity; k++) {
            const int64_t vid = (int64_t)be64(data + index);
            int64_t val = (int64_t)be64(data + index + 8);
            if (vid < 0 || vid >= nvar) { nsk::set_error("graph.factors: variable id outside graph.variables"); return NSK_E_INDEX; }
            if (domain_mask && domain_mask[vid]) {       // value -> dense index, np.searchsorted (213-218)
                const int64_t s = variable[vid].vtf_offset, n = variable[vid].cardinality;
                int64_t lo = 0, hi = n;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) / 2;
                    if (vmap[s + mid].value < val) lo = mid + 1; else hi = mid;
                }
                val = lo;
            }
            fmap[e + k].vid = vid;
            fmap[e + k].dense_equal_to = val;
            index += 16;
        }
        e += arity;
        factor[i].weightId = (int64_t)be64(data + index);
        uint64_t fv = be64(data + index + 8);
        memcpy(&factor[i].featureValue, &fv, 8);
        index += 16;
    }
    return NSK_OK;
}


// graph.domains (dataloading.py:159-187): blocks of (variable id, cardinality, cardinality values),
// all big-endian int64.  Marks the variable, stores its domain in vmap[vtf_offset ..].value and
// rewrites its initialValue as a dense index -- the reference rewrites it EVERY time a domain value
// matches the CURRENT initialValue while scanning j = 0 .. cardinality-1, so a later match can re-map
// an earlier result; restated as coded.
extern "C" int nsk_parse_domains(const uint8_t *data, int64_t nbytes, uint8_t *domain_mask, int64_t nvar,
                                 nsk_variable *variable, nsk_vtf *vmap, int64_t nvtf) {
    int64_t i = 0;
    while (i + 16 <= nbytes) {
        const int64_t vid = (int64_t)be64(data + i), card = (int64_t)be64(data + i + 8);
        i += 16;
        if (vid < 0 || vid >= nvar) { nsk::set_error("graph.domains: variable id outside the variables"); return NSK_E_INDEX; }
        if (card < 0 || i + 8 * card > nbytes) { nsk::set_error("graph.domains truncated"); return NSK_E_INDEX; }
        const int64_t off = variable[vid].vtf_offset;
        if (off < 0 || off + card > nvtf) { nsk::set_error("graph.domains: domain outside vmap"); return NSK_E_INDEX; }
        domain_mask[vid] = 1;
        int64_t init = variable[vid].initialValue;
        for (int64_t j = 0; j < card; j++) {
            const int64_t val = (int64_t)be64(data + i + 8 * j);
            vmap[off + j].value = val;
            if (val == init) init = j;
        }
        variable[vid].initialValue = init;
        i += 8 * card;
    }
    if (i != nbytes) { nsk::set_error("graph.domains: trailing bytes"); return NSK_E_INDEX; }
    return NSK_OK;
}

// dump_probabilities (factorgraph.py:216-229): "<vid> <value> <prob %.3f>" lines; a binary variable
// prints the probability of value 1, any other one line per domain value (vmap.value).
extern "C" int nsk_write_probabilities(const char *path, int64_t nvar, const nsk_variable *variable,
                                       const nsk_vtf *vmap, int64_t nvtf, const int64_t *cstart,
                                       const int64_t *count, int64_t ncount, double epochs) {
    // validate before anything is written: a malformed variable array must not index past vmap / count
    for (int64_t i = 0; i < nvar; i++) {
        const int64_t card = variable[i].cardinality, span = card == 2 ? 1 : card;
        if (card < 1 || cstart[i] < 0 || cstart[i] + span > ncount) {
            nsk::set_error("dump_probabilities: tally slots of a variable lie outside count");
            return NSK_E_INDEX;
        }
        if (card != 2 && (variable[i].vtf_offset < 0 || variable[i].vtf_offset + card > nvtf)) {
            nsk::set_error("dump_probabilities: domain of a variable lies outside vmap");
            return NSK_E_INDEX;
        }
    }
    FILE *f = fopen(path, "w");
    if (!f) { nsk::set_error(std::string("cannot open ") + path); return NSK_E_INVALID; }
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    for (int64_t i = 0; i < nvar; i++) {
        if (variable[i].cardinality == 2) {
            fprintf(f, "%lld %d %.3f\n", (long long)i, 1, (double)count[cstart[i]] / epochs);
            continue;
        }
        for (int64_t k = 0; k < variable[i].cardinality; k++)
            fprintf(f, "%lld %lld %.3f\n", (long long)i, (long long)vmap[variable[i].vtf_offset + k].value,
                    (double)count[cstart[i] + k] / epochs);
    }
    const int bad = ferror(f);
    fclose(f);
    if (bad) { nsk::set_error("write error"); return NSK_E_INVALID; }
    return NSK_OK;
}

// ---- graph-aware partitioning ---------------------------------------------------------
//   nsk_graph_order   <-  salt/src/messages.py:542-590 find_connected_components, :593-670 find_metis_parts
// The reference partitions a factor graph for its minions by connected components or with METIS (objective:
// communication volume) and stores variable -> part.  Here a partition is a VARIABLE ORDER in front of the range
// partition the samplers use (inference.py:17-18): connected components are kept together and, inside a
// component, variables follow a breadth-first (Cuthill-McKee) walk from a pseudo-peripheral variable, so that the
// cut of  [g n / G, (g + 1) n / G)  runs along a few BFS fronts instead of through the caller's id order.  The
// parts are exactly the shard formula's sizes by construction (a METIS part is balanced within a tolerance).
//   method 0: components only (in order of their smallest variable id, ids ascending inside)
//   method 1: components + breadth-first order inside each
//   method 2: components + maximum-adjacency order inside each
//   method 3: ... refined by 20 rounds of median-of-neighbours placement (below)
// order[new id] = old id; cc_id[old id] = component (numbered by smallest member id); *ncc = their number.
extern "C" int nsk_graph_order(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge,
                               const nsk_ftv *fmap, int method, int64_t *order, int64_t *cc_id, int64_t *ncc) {
    if (nvar < 0 || nfactor < 0 || nedge < 0 || !order || (nfactor && !factor) || (nedge && !fmap)) {
        nsk::set_error("nsk_graph_order: bad argument");
        return NSK_E_INVALID;
    }
    // variable -> factors (every edge once; NOOP factors tie nothing together, like remove_noop, messages.py:674)
    std::vector<int64_t> voff((size_t)nvar + 1, 0);
    std::vector<int64_t> fbeg((size_t)nfactor), fend((size_t)nfactor);
    for (int64_t f = 0; f < nfactor; f++) {
        const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
        if (factor[f].arity < 0 || s < 0 || e > nedge) { nsk::set_error("nsk_graph_order: factor members outside fmap"); return NSK_E_INDEX; }
        fbeg[(size_t)f] = s;
        fend[(size_t)f] = factor[f].factorFunction == -1 ? s : e;
        for (int64_t l = s; l < fend[(size_t)f]; l++) {
            if (fmap[l].vid < 0 || fmap[l].vid >= nvar) { nsk::set_error("nsk_graph_order: member outside variables"); return NSK_E_INDEX; }
            voff[(size_t)fmap[l].vid + 1]++;
        }
    }
    for (int64_t v = 0; v < nvar; v++) voff[(size_t)v + 1] += voff[(size_t)v];
    std::vector<int64_t> vfac((size_t)voff[(size_t)nvar]), fill(voff.begin(), voff.end() - 1);
    for (int64_t f = 0; f < nfactor; f++)
        for (int64_t l = fbeg[(size_t)f]; l < fend[(size_t)f]; l++) vfac[(size_t)fill[(size_t)fmap[l].vid]++] = f;
    std::vector<int64_t> comp((size_t)nvar, -1), queue;
    std::vector<uint8_t> fseen((size_t)nfactor, 0);
    queue.reserve((size_t)nvar);
    // breadth-first walk from `start` over unvisited variables (mark = value written into comp[]); appends to
    // `queue` from position q0 on; returns the last variable reached
    auto bfs = [&](int64_t start, int64_t mark, size_t q0, std::vector<int64_t> &touched_f) -> int64_t {
        queue.resize(q0);
        queue.push_back(start);
        comp[(size_t)start] = mark;
        for (size_t h = q0; h < queue.size(); h++) {
            const int64_t v = queue[h];
            for (int64_t j = voff[(size_t)v]; j < voff[(size_t)v + 1]; j++) {
                const int64_t f = vfac[(size_t)j];
                if (fseen[(size_t)f]) continue;             // its members are all queued already
                fseen[(size_t)f] = 1;
                touched_f.push_back(f);
                for (int64_t l = fbeg[(size_t)f]; l < fend[(size_t)f]; l++) {
                    const int64_t u = fmap[l].vid;
                    if (comp[(size_t)u] != mark) { comp[(size_t)u] = mark; queue.push_back(u); }
                }
            }
        }
        return queue.back();
    };
    if (method == 2 || method == 3) {
        // maximum-adjacency order: the next variable is the unvisited one with the most factor links into the visited
        // set (ties: first reached) -- on graphs whose local structure is laced with a few long edges (config #5's
        // 1 % of global members) a breadth-first walk follows the long edges and mixes everything within a few
        // levels, while a variable reached over ONE long edge waits behind the frontier's better-connected ones
        std::vector<int64_t> score((size_t)nvar, 0), ord;
        std::vector<uint8_t> done((size_t)nvar, 0);
        ord.reserve((size_t)nvar);
        typedef std::pair<int64_t, int64_t> Item;                 // (score, -arrival): largest score, earliest arrival first
        std::vector<Item> heap;
        int64_t arrival = 0, ncomp2 = 0;
        for (int64_t s0 = 0; s0 < nvar; s0++) {
            if (done[(size_t)s0]) continue;
            heap.clear();
            heap.emplace_back(0, -(arrival++));
            std::vector<int64_t> who(1, s0);                      // arrival -> variable, per component (offset by first arrival)
            const int64_t a0 = arrival - 1;
            std::push_heap(heap.begin(), heap.end());
            // (lazy deletion: an entry is stale when its score is not the variable's current score)
            std::vector<int64_t> arr_of;                          // unused; kept simple below
            std::vector<std::pair<int64_t, int64_t>> entries;     // (variable, score at push) by arrival - a0
            entries.emplace_back(s0, 0);
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const Item it = heap.back();
                heap.pop_back();
                const std::pair<int64_t, int64_t> &en = entries[(size_t)(-it.second - a0)];
                const int64_t v = en.first;
                if (done[(size_t)v] || en.second != score[(size_t)v]) continue;
                done[(size_t)v] = 1;
                comp[(size_t)v] = ncomp2;
                ord.push_back(v);
                for (int64_t j = voff[(size_t)v]; j < voff[(size_t)v + 1]; j++) {
                    const int64_t f = vfac[(size_t)j];
                    for (int64_t l = fbeg[(size_t)f]; l < fend[(size_t)f]; l++) {
                        const int64_t u = fmap[l].vid;
                        if (done[(size_t)u]) continue;
                        score[(size_t)u]++;
                        entries.emplace_back(u, score[(size_t)u]);
                        heap.emplace_back(score[(size_t)u], -(arrival++));
                        std::push_heap(heap.begin(), heap.end());
                    }
                }
            }
            (void)who; (void)arr_of;
            ncomp2++;
        }
        if (method == 3) {
            // ... refined by rounds of "move every variable half way to the MEDIAN position of its neighbours, then
            // re-rank": the median ignores the few long edges of a variable whose other neighbours sit together, so the
            // order settles into the graph's local (band / mesh) structure -- shuffled 100 000-variable config-#5 graph,
            // 8 parts: communication volume 70 055 after the walk, 32 017 after 20 rounds, 20 233 in the generator's own
            // ids, 453 137 in the shuffled ids.  Components stay together (positions never cross a component: a
            // variable's neighbours are in its own).
            const int rounds = 20;
            std::vector<double> x((size_t)nvar), xn((size_t)nvar);
            for (int64_t i = 0; i < nvar; i++) x[(size_t)ord[(size_t)i]] = (double)i;
            std::vector<int64_t> cstart((size_t)ncomp2 + 1, 0);            // components are contiguous in ord
            for (int64_t v = 0; v < nvar; v++) cstart[(size_t)comp[(size_t)v] + 1]++;
            for (int64_t k = 0; k < ncomp2; k++) cstart[(size_t)k + 1] += cstart[(size_t)k];
            for (int r = 0; r < rounds; r++) {
                nsk::parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
                    std::vector<double> nb;
                    for (int64_t v = b0; v < b1; v++) {
                        nb.clear();
                        for (int64_t j = voff[(size_t)v]; j < voff[(size_t)v + 1]; j++) {
                            const int64_t f = vfac[(size_t)j];
                            for (int64_t l = fbeg[(size_t)f]; l < fend[(size_t)f]; l++)
                                if (fmap[l].vid != v) nb.push_back(x[(size_t)fmap[l].vid]);
                        }
                        double m = x[(size_t)v];
                        if (!nb.empty()) {
                            const size_t k0 = (nb.size() - 1) / 2, k1 = nb.size() / 2;
                            std::nth_element(nb.begin(), nb.begin() + (std::ptrdiff_t)k0, nb.end());
                            const double lo = nb[k0];
                            std::nth_element(nb.begin(), nb.begin() + (std::ptrdiff_t)k1, nb.end());
                            m = 0.5 * (lo + nb[k1]);
                        }
                        xn[(size_t)v] = 0.5 * x[(size_t)v] + 0.5 * m;
                    }
                }, 1024);
                // re-rank inside every component (ties: the previous order)
                nsk::parallel_for(ncomp2, [&](int64_t k0, int64_t k1, int) {
                    for (int64_t k = k0; k < k1; k++)
                        std::stable_sort(ord.begin() + (std::ptrdiff_t)cstart[(size_t)k], ord.begin() + (std::ptrdiff_t)cstart[(size_t)k + 1],
                                         [&](int64_t a, int64_t b) { return xn[(size_t)a] < xn[(size_t)b]; });
                }, 1);
                for (int64_t i = 0; i < nvar; i++) x[(size_t)ord[(size_t)i]] = (double)i;
            }
        }
        for (int64_t i = 0; i < nvar; i++) order[i] = ord[(size_t)i];
        if (cc_id) for (int64_t v = 0; v < nvar; v++) cc_id[v] = comp[(size_t)v];
        if (ncc) *ncc = ncomp2;
        return NSK_OK;
    }
    int64_t ncomp = 0;
    std::vector<int64_t> touched;
    size_t out = 0;
    for (int64_t s0 = 0; s0 < nvar; s0++) {
        if (comp[(size_t)s0] >= 0) continue;
        touched.clear();
        // first walk: the component (marked -2 - ncomp so that a second walk can re-mark it)
        const int64_t far = bfs(s0, -2 - ncomp, out, touched);
        const size_t csize = queue.size() - out;
        if (method == 1 && csize > 2) {                     // second walk from the far end: a pseudo-peripheral start
            for (int64_t f : touched) fseen[(size_t)f] = 0;
            touched.clear();
            (void)bfs(far, ncomp, out, touched);
        } else {
            for (size_t i = out; i < queue.size(); i++) comp[(size_t)queue[i]] = ncomp;
            if (method == 0) std::sort(queue.begin() + (std::ptrdiff_t)out, queue.end());
        }
        out = queue.size();
        ncomp++;
    }
    for (int64_t i = 0; i < nvar; i++) order[i] = queue[(size_t)i];
    if (cc_id) for (int64_t v = 0; v < nvar; v++) cc_id[v] = comp[(size_t)v];
    if (ncc) *ncc = ncomp;
    return NSK_OK;
}

// Communication volume of the range partition into `nparts` shards (what METIS' objtype = vol minimises,
// messages.py:612-615): the number of (variable, foreign part) pairs such that a factor with a member in the
// foreign part reads the variable -- the ghosts all shards hold together, = the values one exchange moves.
// new_id: the position of every variable in the order the partition cuts (NULL: the caller's ids).
extern "C" int nsk_comm_volume(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge,
                               const nsk_ftv *fmap, const int64_t *new_id, int nparts, int64_t *volume) {
    if (nparts < 1 || nparts > 64 || !volume) { nsk::set_error("nsk_comm_volume: 1 .. 64 parts"); return NSK_E_INVALID; }
    std::vector<uint64_t> need((size_t)nvar, 0);
    auto part_of = [&](int64_t v) -> int {              // the part whose range [g n / G, (g + 1) n / G) holds id v
        const int64_t id = new_id ? new_id[v] : v;
        int g = (int)(((__int128)id * nparts + nparts - 1) / std::max<int64_t>(nvar, 1));
        g = std::min(std::max(g, 0), nparts - 1);
        while (g > 0 && (int64_t)(((__int128)g * nvar) / nparts) > id) g--;
        while (g + 1 < nparts && (int64_t)(((__int128)(g + 1) * nvar) / nparts) <= id) g++;
        return g;
    };
    for (int64_t f = 0; f < nfactor; f++) {
        if (factor[f].factorFunction == -1) continue;
        const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
        if (factor[f].arity < 0 || s < 0 || e > nedge) { nsk::set_error("nsk_comm_volume: factor members outside fmap"); return NSK_E_INDEX; }
        uint64_t parts = 0;
        for (int64_t l = s; l < e; l++) {
            if (fmap[l].vid < 0 || fmap[l].vid >= nvar) { nsk::set_error("nsk_comm_volume: member outside variables"); return NSK_E_INDEX; }
            parts |= 1ull << part_of(fmap[l].vid);
        }
        if (parts & (parts - 1))                            // the factor spans parts: every member is read by the others
            for (int64_t l = s; l < e; l++) need[(size_t)fmap[l].vid] |= parts & ~(1ull << part_of(fmap[l].vid));
    }
    int64_t vol = 0;
    for (int64_t v = 0; v < nvar; v++) vol += __builtin_popcountll(need[(size_t)v]);
    *volume = vol;
    return NSK_OK;
}
