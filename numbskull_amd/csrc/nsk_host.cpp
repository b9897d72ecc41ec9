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
    // lengths over EVERY edge (34-38), then exclusive prefix (41-46)
    for (int64_t l = 0; l < nedge; l++) {
        int64_t idx;
        if (!slot_of(l, idx)) { nsk::set_error("compute_var_map: edge refers outside variables/vmap"); return NSK_E_INDEX; }
        vmap[idx].factor_index_length += 1;
    }
    int64_t last_len = 0, last_off = 0;
    for (int64_t i = 0; i < nvtf; i++) {
        vmap[i].factor_index_offset = last_off + last_len;
        last_len = vmap[i].factor_index_length;
        last_off = vmap[i].factor_index_offset;
    }
    // scatter factor ids in factor order, skipping factors_to_skip (49-65)
    std::vector<int64_t> cursor((size_t)nvtf);
    for (int64_t i = 0; i < nvtf; i++) cursor[i] = vmap[i].factor_index_offset;
    int64_t fts = 0;
    for (int64_t f = 0; f < nfactor; f++) {
        if (fts < nskip && factors_to_skip[fts] == f) { fts++; continue; }
        const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
        if (s < 0 || e > nedge) { nsk::set_error("compute_var_map: factor members outside fmap"); return NSK_E_INDEX; }
        for (int64_t l = s; l < e; l++) {
            int64_t idx;
            slot_of(l, idx);
            if (cursor[idx] >= nfi) { nsk::set_error("compute_var_map: index out of bounds for factor_index (IndexError in the reference)"); return NSK_E_INDEX; }
            factor_index[cursor[idx]++] = f;
        }
    }
    // per slot: sort, drop duplicates in place, shrink the length (68-81)
    for (int64_t i = 0; i < nvtf; i++) {
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
