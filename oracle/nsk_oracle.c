/*
 * nsk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see nsk_oracle.h).
 *
 * Restates, function by function, the reference hot path of HazyResearch/numbskull:
 *   numbskull/inference.py   gibbsthread 10-33, draw_sample 36-52, potential 55-71,
 *                            eval_factor 149-413
 *   numbskull/learning.py    learnthread 12-31, get_factor_id_range 34-43,
 *                            sample_and_sgd 46-125
 *   numbskull/dataloading.py compute_var_map 16-81
 * plus the two generators the reference draws from in pure-Python mode (numpy legacy
 * MT19937 for np.random.rand(), CPython's MT19937 for random.random()).
 *
 * Two modes:
 *   *_ref  : sequential scan + MT19937 + libm exp  == the reference, bit for bit
 *            (pinned by tests/golden/ fixtures captured from the reference itself).
 *   *_dev  : the same per-variable rule, visited phase by phase in a caller-given order,
 *            with Philox4x32-10 uniforms and the deterministic exp -- the semantics the
 *            HIP kernels implement (DESIGN.md "device mode").  HIP == *_dev bit for bit.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "nsk_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * MT19937
 * ---------------------------------------------------------------------------------------- */
static void mt_init_genrand(orc_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

/* np.random.seed(int): numpy's legacy seeding is Knuth's init_genrand. */
void orc_mt_seed_numpy(orc_mt *s, uint32_t seed) { mt_init_genrand(s, seed); }

/* random.seed(int): CPython feeds |seed| as little-endian 32-bit words to init_by_array. */
void orc_mt_seed_python(orc_mt *s, uint64_t seed) {
    uint32_t key[2];
    int keylen = 1;
    key[0] = (uint32_t)(seed & 0xffffffffu);
    key[1] = (uint32_t)(seed >> 32);
    if (key[1]) keylen = 2;
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = 624 > keylen ? 624 : keylen;
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= keylen) j = 0;
    }
    for (k = 623; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = 624;
}

static uint32_t mt_next(orc_mt *s) {
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double orc_u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

double orc_mt_res53(orc_mt *s) {
    uint32_t a = mt_next(s);
    uint32_t b = mt_next(s);
    return orc_u53(a, b);
}

/* ------------------------------------------------------------------------------------------
 * Deterministic exp (specification shared with the device code, DESIGN.md "nsk_exp"):
 * k = rint(x/ln2); r = x - k*ln2 (two-term Cody-Waite, fma); degree-13 Taylor in Horner
 * form with fma; result scaled by 2^k in two exact power-of-two steps.
 * ---------------------------------------------------------------------------------------- */
static double pow2i(int k) {
    union { uint64_t u; double d; } v;
    v.u = (uint64_t)(k + 1023) << 52;
    return v.d;
}

double orc_exp_det(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019412) return 0.0;
    const double INV_LN2 = 1.4426950408889634;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = rint(x * INV_LN2);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;            /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);         /* 1/12! */
    p = fma(p, r, 1.0 / 39916800.0);          /* 1/11! */
    p = fma(p, r, 1.0 / 3628800.0);           /* 1/10! */
    p = fma(p, r, 1.0 / 362880.0);            /* 1/9!  */
    p = fma(p, r, 1.0 / 40320.0);             /* 1/8!  */
    p = fma(p, r, 1.0 / 5040.0);              /* 1/7!  */
    p = fma(p, r, 1.0 / 720.0);               /* 1/6!  */
    p = fma(p, r, 1.0 / 120.0);               /* 1/5!  */
    p = fma(p, r, 1.0 / 24.0);                /* 1/4!  */
    p = fma(p, r, 1.0 / 6.0);                 /* 1/3!  */
    p = fma(p, r, 0.5);                       /* 1/2!  */
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    int k1 = k / 2;
    int k2 = k - k1;
    return (p * pow2i(k1)) * pow2i(k2);
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11), counter-based generator of the device mode.
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]) {
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------------------------------
 * eval_factor (inference.py:149-413).  Values are written to *out as double (every function
 * returns a small integer except RATIO).  Returns ORC_OK / ORC_E_*.
 * ---------------------------------------------------------------------------------------- */
#define F_NOOP (-1)
#define F_IMPLY_NATURAL 0
#define F_OR 1
#define F_AND 2
#define F_EQUAL 3
#define F_ISTRUE 4
#define F_LINEAR 7
#define F_RATIO 8
#define F_LOGICAL 9
#define F_AND_CAT 12
#define F_IMPLY_MLN 13
#define F_OR_CAT 14
#define F_EQUAL_CAT_CONST 15
#define F_IMPLY_NATURAL_CAT 16
#define F_IMPLY_MLN_CAT 17
#define F_DP_GEN_CLASS_PRIOR 18
#define F_DP_GEN_LF_PRIOR 19
#define F_DP_GEN_LF_PROPENSITY 20
#define F_DP_GEN_LF_ACCURACY 21
#define F_DP_GEN_LF_CLASS_PROPENSITY 22
#define F_DP_GEN_DEP_FIXING 23
#define F_DP_GEN_DEP_REINFORCING 24
#define F_DP_GEN_DEP_EXCLUSIVE 25
#define F_DP_GEN_DEP_SIMILAR 26
#define F_UFO 30

/* value of member at absolute edge index l: the candidate if it is the sampled variable */
#define EDGE_OK(l) ((l) >= 0 && (l) < g->nedge)
#define MEMBER(l) ((g->fmap[(l)].vid == var_samp) ? value : var_value[g->fmap[(l)].vid])

/* head lookup of IMPLY_MLN / IMPLY_NATURAL_CAT / IMPLY_MLN_CAT: the reference reads
 * var_value[var_copy][l] with l the ABSOLUTE EDGE INDEX (inference.py:243,277,292). */
static int head_value(const orc_graph *g, int64_t l, int64_t var_samp, int64_t value,
                      const int64_t *var_value, int64_t *head) {
    if (g->fmap[l].vid == var_samp) { *head = value; return ORC_OK; }
    if (g->head_by_vid) { *head = var_value[g->fmap[l].vid]; return ORC_OK; }
    if (l >= g->nvar) return ORC_E_INDEX;   /* IndexError in the reference */
    *head = var_value[l];
    return ORC_OK;
}

int orc_eval_factor(const orc_graph *g, int64_t factor_id, int64_t var_samp, int64_t value,
                    const int64_t *var_value, double *out) {
    const orc_factor *fac = &g->factor[factor_id];
    const int64_t ftv_start = fac->ftv_offset;
    const int64_t ftv_end = ftv_start + fac->arity;
    const int fn = fac->factorFunction;
    int64_t l, v, head;
    int rc;

    if (fn == F_NOOP) { *out = 0; return ORC_OK; }
    /* every other branch touches fmap[ftv_start .. ftv_end) at least */
    if (fac->arity < 0 || ftv_start < 0 || ftv_end > g->nedge) return ORC_E_INDEX;

    switch (fn) {
    case F_IMPLY_NATURAL:                                     /* 162-176 */
        for (l = ftv_start; l < ftv_end; l++) {
            v = MEMBER(l);
            if (v == 0) { *out = 0; return ORC_OK; }
        }
        if (fac->arity < 1) return ORC_E_INDEX;
        head = MEMBER(ftv_end - 1);
        *out = head ? 1 : -1;
        return ORC_OK;
    case F_OR:                                                /* 177-183 */
        for (l = ftv_start; l < ftv_end; l++)
            if (MEMBER(l) == 1) { *out = 1; return ORC_OK; }
        *out = -1;
        return ORC_OK;
    case F_EQUAL:                                             /* 184-192 */
        if (!EDGE_OK(ftv_start)) return ORC_E_INDEX;
        v = MEMBER(ftv_start);
        for (l = ftv_start + 1; l < ftv_end; l++)
            if (v != MEMBER(l)) { *out = -1; return ORC_OK; }
        *out = 1;
        return ORC_OK;
    case F_AND:
    case F_ISTRUE:                                            /* 193-200 */
        for (l = ftv_start; l < ftv_end; l++)
            if (MEMBER(l) == 0) { *out = -1; return ORC_OK; }
        *out = 1;
        return ORC_OK;
    case F_LINEAR:
    case F_RATIO:
    case F_LOGICAL: {                                         /* 201-231 */
        if (fac->arity < 1) return ORC_E_INDEX;
        int64_t res = 0;
        head = MEMBER(ftv_end - 1);
        for (l = ftv_start; l < ftv_end - 1; l++) {
            if (MEMBER(l) == head) {
                if (fn == F_LOGICAL) { *out = 1; return ORC_OK; }
                res++;
            }
        }
        if (fn == F_LINEAR) *out = (double)res;
        else if (fn == F_RATIO) *out = log((double)(res + 1));
        else *out = 0;
        return ORC_OK;
    }
    case F_IMPLY_MLN:                                         /* 232-246 */
        if (fac->arity < 1) return ORC_E_INDEX;
        for (l = ftv_start; l < ftv_end - 1; l++)
            if (MEMBER(l) == 0) { *out = 1; return ORC_OK; }
        if ((rc = head_value(g, ftv_end - 1, var_samp, value, var_value, &head))) return rc;
        *out = head ? 1 : 0;
        return ORC_OK;
    case F_AND_CAT:
    case F_EQUAL_CAT_CONST:                                   /* 251-258 */
        for (l = ftv_start; l < ftv_end; l++)
            if (MEMBER(l) != g->fmap[l].dense_equal_to) { *out = 0; return ORC_OK; }
        *out = 1;
        return ORC_OK;
    case F_OR_CAT:                                            /* 259-265 */
        for (l = ftv_start; l < ftv_end; l++)
            if (MEMBER(l) == g->fmap[l].dense_equal_to) { *out = 1; return ORC_OK; }
        *out = -1;
        return ORC_OK;
    case F_IMPLY_NATURAL_CAT:                                 /* 266-280 */
        if (fac->arity < 1) return ORC_E_INDEX;
        for (l = ftv_start; l < ftv_end - 1; l++)
            if (MEMBER(l) != g->fmap[l].dense_equal_to) { *out = 0; return ORC_OK; }
        l = ftv_end - 1;
        if ((rc = head_value(g, l, var_samp, value, var_value, &head))) return rc;
        *out = (head == g->fmap[l].dense_equal_to) ? 1 : -1;
        return ORC_OK;
    case F_IMPLY_MLN_CAT:                                     /* 281-295 */
        if (fac->arity < 1) return ORC_E_INDEX;
        for (l = ftv_start; l < ftv_end - 1; l++)
            if (MEMBER(l) != g->fmap[l].dense_equal_to) { *out = 1; return ORC_OK; }
        l = ftv_end - 1;
        if ((rc = head_value(g, l, var_samp, value, var_value, &head))) return rc;
        *out = (head == g->fmap[l].dense_equal_to) ? 1 : 0;
        return ORC_OK;
    default:
        break;
    }

    /* data-programming functions and UFO read fixed member positions */
    if (fn >= F_DP_GEN_CLASS_PRIOR && fn <= F_DP_GEN_DEP_SIMILAR) {
        int need = (fn <= F_DP_GEN_LF_PROPENSITY) ? 1
                 : (fn == F_DP_GEN_DEP_FIXING || fn == F_DP_GEN_DEP_REINFORCING) ? 3 : 2;
        if (!EDGE_OK(ftv_start + need - 1)) return ORC_E_INDEX;
        int64_t x0 = MEMBER(ftv_start);
        int64_t x1 = need >= 2 ? MEMBER(ftv_start + 1) : 0;
        int64_t x2 = need >= 3 ? MEMBER(ftv_start + 2) : 0;
        int64_t abstain;
        switch (fn) {
        case F_DP_GEN_CLASS_PRIOR:                            /* 301-305 */
            *out = (x0 == 1) ? 1 : -1;
            return ORC_OK;
        case F_DP_GEN_LF_PRIOR:                               /* 306-315 */
            *out = (x0 == 2) ? -1 : (x0 == 0 ? 0 : 1);
            return ORC_OK;
        case F_DP_GEN_LF_PROPENSITY:                          /* 316-320 */
            abstain = g->variable[g->fmap[ftv_start].vid].cardinality - 1;
            *out = (x0 == abstain) ? 0 : 1;
            return ORC_OK;
        case F_DP_GEN_LF_ACCURACY:                            /* 321-332 */
            abstain = g->variable[g->fmap[ftv_start + 1].vid].cardinality - 1;
            *out = (x1 == abstain) ? 0 : (x0 == x1 ? 1 : -1);
            return ORC_OK;
        case F_DP_GEN_LF_CLASS_PROPENSITY:                    /* 333-346 */
            abstain = g->variable[g->fmap[ftv_start + 1].vid].cardinality - 1;
            *out = (x1 == abstain) ? 0 : (x0 == 1 ? 1 : -1);
            return ORC_OK;
        case F_DP_GEN_DEP_FIXING:                             /* 347-363 */
            abstain = g->variable[g->fmap[ftv_start + 1].vid].cardinality - 1;
            if (x1 == abstain) *out = (x2 != 1) ? -1 : 0;
            else if (x1 == 0 && x2 == 1 && x0 == 1) *out = 1;
            else if (x1 == 1 && x2 == 0 && x0 == 0) *out = 1;
            else *out = 0;
            return ORC_OK;
        case F_DP_GEN_DEP_REINFORCING:                        /* 364-380 */
            abstain = g->variable[g->fmap[ftv_start + 1].vid].cardinality - 1;
            if (x1 == abstain) *out = (x2 != 1) ? -1 : 0;
            else if (x1 == 0 && x2 == 0 && x0 == 0) *out = 1;
            else if (x1 == 1 && x2 == 1 && x0 == 1) *out = 1;
            else *out = 0;
            return ORC_OK;
        case F_DP_GEN_DEP_EXCLUSIVE:                          /* 381-387 */
            abstain = g->variable[g->fmap[ftv_start].vid].cardinality - 1;
            *out = (x0 == abstain || x1 == abstain) ? 0 : -1;
            return ORC_OK;
        case F_DP_GEN_DEP_SIMILAR:                            /* 388-393 */
            *out = (x0 == x1) ? 1 : 0;
            return ORC_OK;
        }
    }
    if (fn == F_UFO) {                                        /* 398-405 */
        if (!EDGE_OK(ftv_start)) return ORC_E_INDEX;
        v = MEMBER(ftv_start);
        if (v == 0) { *out = 0; return ORC_OK; }
        l = ftv_start + v - 1;
        if (!EDGE_OK(l)) return ORC_E_INDEX;
        *out = (double)MEMBER(l);
        return ORC_OK;
    }
    return ORC_E_FACTOR_FUNC;                                 /* 410-413 */
}

/* potential (inference.py:55-71): sum in factor_index order, product then add (no fma) */
static int potential_impl(const orc_graph *g, int64_t var_samp, int64_t value,
                          const int64_t *var_value, const double *weight_value, double *out) {
    const orc_variable *var = &g->variable[var_samp];
    int64_t varval_off = (var->dataType == 0) ? 0 : value;
    const orc_vtf *vtf = &g->vmap[var->vtf_offset + varval_off];
    int64_t start = vtf->factor_index_offset;
    int64_t end = start + vtf->factor_index_length;
    double p = 0.0;
    for (int64_t k = start; k < end; k++) {
        int64_t fid = g->factor_index[k];
        double e;
        int rc = orc_eval_factor(g, fid, var_samp, value, var_value, &e);
        if (rc) return rc;
        double t = weight_value[g->factor[fid].weightId] * e;
        p = p + t;
    }
    *out = p;
    return ORC_OK;
}

int orc_potential(const orc_graph *g, int64_t var_samp, int64_t value, const int64_t *var_value,
                  const double *weight_value, double *out) {
    return potential_impl(g, var_samp, value, var_value, weight_value, out);
}

/* draw_sample (inference.py:36-52), split at the point where the reference draws its uniform:
 * fill_Z = lines 39-47 (potentials, exp, running sum); pick = lines 50-52. */
static int fill_Z(const orc_graph *g, int64_t var_samp, double *Z, const int64_t *var_value,
                  const double *weight_value, int det) {
    int64_t card = g->variable[var_samp].cardinality;
    if (card < 1) return ORC_E_INDEX;
    for (int64_t value = 0; value < card; value++) {
        double p;
        int rc = potential_impl(g, var_samp, value, var_value, weight_value, &p);
        if (rc) return rc;
        Z[value] = det ? orc_exp_det(p) : exp(p);
    }
    for (int64_t j = 1; j < card; j++) Z[j] += Z[j - 1];
    return ORC_OK;
}

static int64_t pick(const double *Z, int64_t card, double u) {
    double z = u * Z[card - 1];
    for (int64_t j = 0; j < card; j++)
        if (Z[j] >= z) return j;
    return 0;                                      /* np.argmax of an all-False mask is 0 */
}

static int draw_sample(const orc_graph *g, int64_t var_samp, double *Z, const int64_t *var_value,
                       const double *weight_value, double u, int det, int64_t *result) {
    int rc = fill_Z(g, var_samp, Z, var_value, weight_value, det);
    if (rc) return rc;
    *result = pick(Z, g->variable[var_samp].cardinality, u);
    return ORC_OK;
}

/* gibbsthread (inference.py:10-33), one shard, reference mode */
int orc_gibbs_shard_ref(const orc_graph *g, int64_t shardID, int64_t nshards, double *Z,
                        const int64_t *cstart, int64_t *count, int64_t *var_value,
                        const double *weight_value, int sample_evidence, int burnin,
                        orc_mt *np_rng) {
    int64_t nvar = g->nvar;
    int64_t start = (shardID * nvar) / nshards;
    int64_t end = ((shardID + 1) * nvar) / nshards;
    for (int64_t var_samp = start; var_samp < end; var_samp++) {
        const orc_variable *var = &g->variable[var_samp];
        if (var->isEvidence == 4) continue;
        if (var->isEvidence == 0 || sample_evidence) {
            int rc = fill_Z(g, var_samp, Z, var_value, weight_value, 0);
            if (rc) return rc;
            int64_t v = pick(Z, var->cardinality, orc_mt_res53(np_rng));
            var_value[var_samp] = v;
            if (!burnin) {
                if (var->cardinality == 2) count[cstart[var_samp]] += v;
                else count[cstart[var_samp] + v] += 1;
            }
        }
    }
    return ORC_OK;
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

static void factor_id_range(const orc_graph *g, int64_t var_samp, int64_t val, int64_t *start,
                            int64_t *end) {                   /* learning.py:34-43 */
    const orc_variable *var = &g->variable[var_samp];
    int64_t off = (var->dataType == 0) ? 0 : val;
    const orc_vtf *vtf = &g->vmap[var->vtf_offset + off];
    *start = vtf->factor_index_offset;
    *end = *start + vtf->factor_index_length;
}

/* sample_and_sgd (learning.py:46-125), reference mode: per-visit weight update */
static int sample_and_sgd_ref(const orc_graph *g, int64_t var_samp, double step, int regularization,
                              double reg_param, int64_t truncation, double *Z, int64_t *fids,
                              int64_t *var_value, int64_t *var_value_evid, double *weight_value,
                              int learn_non_evidence, orc_mt *np_rng, orc_mt *py_rng) {
    const orc_variable *var = &g->variable[var_samp];
    int64_t evidence, proposal;
    int rc;
    if (var->isEvidence != 1) {
        rc = fill_Z(g, var_samp, Z, var_value_evid, weight_value, 0);
        if (rc) return rc;
        evidence = pick(Z, var->cardinality, orc_mt_res53(np_rng));
    } else {
        evidence = var->initialValue;
    }
    var_value_evid[var_samp] = evidence;
    rc = fill_Z(g, var_samp, Z, var_value, weight_value, 0);
    if (rc) return rc;
    proposal = pick(Z, var->cardinality, orc_mt_res53(np_rng));
    var_value[var_samp] = proposal;
    if (!learn_non_evidence && var->isEvidence != 1) return ORC_OK;

    int64_t s0, e0, s = 0;
    factor_id_range(g, var_samp, evidence, &s0, &e0);
    if (evidence != proposal) {
        int64_t s1, e1;
        factor_id_range(g, var_samp, proposal, &s1, &e1);
        for (int64_t k = s0; k < e0; k++) fids[s++] = g->factor_index[k];
        for (int64_t k = s1; k < e1; k++) fids[s++] = g->factor_index[k];
        qsort(fids, (size_t)s, sizeof(int64_t), cmp_i64);
    } else {
        for (int64_t k = s0; k < e0; k++) fids[s++] = g->factor_index[k];
    }

    int truncate = 0;
    if (regularization == 1) truncate = orc_mt_res53(py_rng) < 1.0 / (double)truncation;

    int64_t last_fid = -1;
    for (int64_t i = 0; i < s; i++) {
        int64_t fid = fids[i];
        if (fid == last_fid) continue;
        last_fid = fid;
        int64_t wid = g->factor[fid].weightId;
        if (g->weight[wid].isFixed) continue;
        double p0, p1;
        rc = orc_eval_factor(g, fid, var_samp, evidence, var_value_evid, &p0);
        if (rc) return rc;
        rc = orc_eval_factor(g, fid, var_samp, proposal, var_value, &p1);
        if (rc) return rc;
        double gradient = (p1 - p0) * g->factor[fid].featureValue;
        double w = weight_value[wid];
        if (regularization == 2) {
            w *= (1.0 / (1.0 + reg_param * step));
            w -= step * gradient;
        } else if (regularization == 1) {
            w -= step * gradient;
            if (truncate) {
                double l1delta = reg_param * step * (double)truncation;
                w = (w > 0) ? fmax(0.0, w - l1delta) : fmin(0.0, w + l1delta);
            }
        } else {
            w -= step * gradient;
        }
        weight_value[wid] = w;
    }
    return ORC_OK;
}

int orc_learn_shard_ref(const orc_graph *g, int64_t shardID, int64_t nshards, double step,
                        int regularization, double reg_param, int64_t truncation, double *Z,
                        int64_t *fids, int64_t *var_value, int64_t *var_value_evid,
                        double *weight_value, int learn_non_evidence, orc_mt *np_rng,
                        orc_mt *py_rng) {
    int64_t nvar = g->nvar;
    int64_t start = (shardID * nvar) / nshards;
    int64_t end = ((shardID + 1) * nvar) / nshards;
    for (int64_t var_samp = start; var_samp < end; var_samp++) {
        if (g->variable[var_samp].isEvidence == 4) continue;
        int rc = sample_and_sgd_ref(g, var_samp, step, regularization, reg_param, truncation, Z,
                                    fids, var_value, var_value_evid, weight_value,
                                    learn_non_evidence, np_rng, py_rng);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Checker of a chromatic colouring (test infrastructure for the device mode): a colour class is
 * sampled in parallel, so no sampled variable may READ another variable of its own colour.
 * reads(v) = the members of every factor in v's lists -- as eval_factor walks them: the arity,
 * the fixed member positions of the data-programming functions (inference.py:301-393), UFO's
 * value-indexed member (398-405) -- plus, with the literal head lookup (243, 277, 292), the
 * variable whose id equals the head's absolute edge index.  Returns -1 when the colouring is
 * valid, else the id of a variable that reads a same-coloured one (*other = that one).
 * ---------------------------------------------------------------------------------------- */
int64_t orc_check_coloring(const orc_graph *g, const int32_t *color, int64_t *other) {
    for (int64_t v = 0; v < g->nvar; v++) {
        if (color[v] < 0) continue;
        const orc_variable *var = &g->variable[v];
        const int64_t nslots = var->dataType == 0 ? 1 : var->cardinality;
        for (int64_t k = 0; k < nslots; k++) {
            const orc_vtf *vt = &g->vmap[var->vtf_offset + k];
            for (int64_t j = 0; j < vt->factor_index_length; j++) {
                const orc_factor *fa = &g->factor[g->factor_index[vt->factor_index_offset + j]];
                const int fn = fa->factorFunction;
                if (fn == -1) continue;
                int64_t need = (fn == 21 || fn == 22 || fn == 25 || fn == 26) ? 2
                             : (fn == 23 || fn == 24) ? 3 : (fn >= 18 && fn <= 20) ? 1 : 0;
                int64_t s = fa->ftv_offset, e = s + (fa->arity > need ? fa->arity : need);
                if (fn == 30 && s >= 0 && s < g->nedge) {
                    int64_t r = s + g->variable[g->fmap[s].vid].cardinality - 1;
                    if (r > e) e = r;
                }
                if (e > g->nedge) e = g->nedge;
                for (int64_t l = s; l < e; l++) {
                    const int64_t b = g->fmap[l].vid;
                    if (b != v && b >= 0 && b < g->nvar && color[b] == color[v]) { if (other) *other = b; return v; }
                }
                if ((fn == 13 || fn == 16 || fn == 17) && !g->head_by_vid) {
                    const int64_t b = s + fa->arity - 1;
                    if (b != v && b >= 0 && b < g->nvar && color[b] == color[v]) { if (other) *other = b; return v; }
                }
            }
        }
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Device mode: Philox uniforms keyed (seed; vid, stream, sweep), orc_exp_det,
 * phases = colour classes, weights frozen inside a phase and updated at its end from
 * order-independent fixed-point gradient sums.
 * ---------------------------------------------------------------------------------------- */
static int64_t max_card(const orc_graph *g) {
    int64_t m = 1;
    for (int64_t i = 0; i < g->nvar; i++)
        if (g->variable[i].cardinality > m) m = g->variable[i].cardinality;
    return m;
}

/* The variables of one colour class do not read each other (orc_check_coloring): the device-mode sweeps may walk a
 * class with several threads -- same values, tallies and (integer) gradient sums whatever their number.  Test speed
 * only: the walks of the 10M / 50M graphs dominated the GPU suite.  Default 1. */
static int orc_dev_threads = 1;
void orc_set_threads(int n) { orc_dev_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
typedef struct {
    int (*fn)(void *ctx, int64_t i0, int64_t i1);
    void *ctx;
    int64_t i0, i1;
    int rc;
} orc_par_arg;
static void *orc_par_worker(void *p) { orc_par_arg *a = (orc_par_arg *)p; a->rc = a->fn(a->ctx, a->i0, a->i1); return NULL; }
/* fn(ctx, i0, i1) over [b, e) in contiguous chunks; the first error code in index order */
static int orc_parallel_range(int64_t b, int64_t e, int (*fn)(void *, int64_t, int64_t), void *ctx) {
    int T = orc_dev_threads;
    if (e - b < 4096 || T <= 1) return fn(ctx, b, e);
    if ((int64_t)T > (e - b) / 1024) T = (int)((e - b) / 1024);
    pthread_t th[256];
    orc_par_arg args[256];
    for (int t = 0; t < T; t++) {
        args[t].fn = fn; args[t].ctx = ctx; args[t].rc = ORC_OK;
        args[t].i0 = b + (e - b) * t / T; args[t].i1 = b + (e - b) * (t + 1) / T;
        pthread_create(&th[t], NULL, orc_par_worker, &args[t]);
    }
    int rc = ORC_OK;
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); if (!rc) rc = args[t].rc; }
    return rc;
}

typedef struct {
    const orc_graph *g; const int64_t *order; const int64_t *cstart; int64_t *count; int64_t *var_value;
    const double *weight_value; int sample_evidence, burnin; uint64_t seed, sweep;
} orc_gibbs_ctx;
static int orc_gibbs_range_dev(void *vctx, int64_t i0, int64_t i1);

int orc_gibbs_sweep_dev(const orc_graph *g, const int64_t *order, const int64_t *phase_start,
                        int64_t nphase, const int64_t *cstart, int64_t *count, int64_t *var_value,
                        const double *weight_value, int sample_evidence, int burnin,
                        uint64_t seed, uint64_t sweep) {
    orc_gibbs_ctx ctx = {g, order, cstart, count, var_value, weight_value, sample_evidence, burnin, seed, sweep};
    int rc = ORC_OK;
    for (int64_t p = 0; p < nphase && !rc; p++)
        rc = orc_parallel_range(phase_start[p], phase_start[p + 1], orc_gibbs_range_dev, &ctx);
    return rc;
}

static int orc_gibbs_range_dev(void *vctx, int64_t i0, int64_t i1) {
    const orc_gibbs_ctx *x = (const orc_gibbs_ctx *)vctx;
    const orc_graph *g = x->g;
    const int64_t *order = x->order, *cstart = x->cstart;
    int64_t *count = x->count, *var_value = x->var_value;
    const double *weight_value = x->weight_value;
    const int sample_evidence = x->sample_evidence, burnin = x->burnin;
    const uint64_t seed = x->seed, sweep = x->sweep;
    double *Z = (double *)malloc(sizeof(double) * (size_t)max_card(g));
    int rc = ORC_OK;
    {
        for (int64_t i = i0; i < i1; i++) {
            int64_t v = order[i];
            const orc_variable *var = &g->variable[v];
            if (var->isEvidence == 4) continue;
            if (!(var->isEvidence == 0 || sample_evidence)) continue;
            /* device-mode generator of the inference sweep: the variable's
             * generator id q is its position in the library's layout (bits 0-39 of g->rng_id; identity
             * when absent).  Pair scheme (bit 40 clear): ids q and q + 64 with equal q >> 7 share ONE
             * Philox block -- counter ((q >> 7) * 64 + (q & 63), 0, sweep), words 0-1 for the lower id,
             * 2-3 for the upper.  Quad scheme (bit 40 set: positions inside segments with draw tables):
             * ids q, q + 64, q + 128, q + 192 with equal q >> 8 share TWO blocks -- counter
             * ((q >> 8) * 64 + (q & 63), stream, sweep), stream 2 word (q >> 6) & 3 = the high word,
             * stream 3 the same word = the low word */
            uint32_t r[4], a, b;
            const uint64_t gid = g->rng_id ? (uint64_t)g->rng_id[v] : (uint64_t)v;
            const uint64_t q = gid & 0xFFFFFFFFFFull;
            if ((gid >> 40) & 3u) {
                /* bit 41: the WIDE scheme (positions inside wide quads: one lane of the library's kernel samples four
                 * consecutive positions) -- the same two blocks per quad and lane dealt the other way round: ids
                 * 4 i .. 4 i + 3 share counter ((q >> 8) * 64 + ((q >> 2) & 63), stream, sweep), word q & 3 */
                const int wide = (int)((gid >> 41) & 1u);
                const uint32_t c0 = wide ? (uint32_t)(((q >> 8) << 6) | ((q >> 2) & 63u)) : (uint32_t)(((q >> 8) << 6) | (q & 63u));
                const uint32_t j = wide ? (uint32_t)(q & 3u) : (uint32_t)((q >> 6) & 3u);
                orc_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c0, 2u,
                               (uint32_t)sweep, (uint32_t)(sweep >> 32) ^ g->rng_tag, r);
                a = r[j];
                orc_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c0, 3u,
                               (uint32_t)sweep, (uint32_t)(sweep >> 32) ^ g->rng_tag, r);
                b = r[j];
            } else {
                orc_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32),
                               (uint32_t)(((q >> 7) << 6) | (q & 63u)), 0u,
                               (uint32_t)sweep, (uint32_t)(sweep >> 32) ^ g->rng_tag, r);
                const int half = (int)((q >> 6) & 1u);
                a = half ? r[2] : r[0];
                b = half ? r[3] : r[1];
            }
            int64_t nv;
            rc = draw_sample(g, v, Z, var_value, weight_value, orc_u53(a, b), 1, &nv);
            if (rc) break;
            var_value[v] = nv;
            if (!burnin) {
                if (var->cardinality == 2) count[cstart[v]] += nv;
                else count[cstart[v] + nv] += 1;
            }
        }
    }
    free(Z);
    return rc;
}

static double powi_det(double a, uint64_t k) {
    double r = 1.0, b = a;
    while (k) {
        if (k & 1) r *= b;
        k >>= 1;
        if (k) b *= b;
    }
    return r;
}

/* gradient of one (variable, factor) visit into the fixed-point accumulators */
static int accumulate_visit(const orc_graph *g, int64_t fid, int64_t v, int64_t evidence,
                            int64_t proposal, const int64_t *var_value,
                            const int64_t *var_value_evid, int truncate, int64_t *G, int64_t *K,
                            int64_t *T) {
    int64_t wid = g->factor[fid].weightId;
    if (g->weight[wid].isFixed) return ORC_OK;
    double p0, p1;
    int rc = orc_eval_factor(g, fid, v, evidence, var_value_evid, &p0);
    if (rc) return rc;
    rc = orc_eval_factor(g, fid, v, proposal, var_value, &p1);
    if (rc) return rc;
    double gradient = (p1 - p0) * g->factor[fid].featureValue;
    /* (integer sums: the order of the adds does not matter, so the threads of a colour class -- orc_set_threads -- add
     * atomically into the same arrays) */
    __atomic_fetch_add(&G[wid], (int64_t)llrint(gradient * ldexp(1.0, 32 - g->grad_shift)), __ATOMIC_RELAXED);
    __atomic_fetch_add(&K[wid], (int64_t)1, __ATOMIC_RELAXED);
    if (truncate) __atomic_fetch_add(&T[wid], (int64_t)1, __ATOMIC_RELAXED);
    return ORC_OK;
}

typedef struct {
    const orc_graph *g; const int64_t *order; int regularization; int64_t truncation; int64_t *var_value, *var_value_evid;
    const double *weight_samp; int learn_non_evidence; uint64_t seed, sweep; int64_t *G, *K, *T;
} orc_learn_ctx;
static int orc_learn_range_dev(void *vctx, int64_t i0, int64_t i1) {
    const orc_learn_ctx *x = (const orc_learn_ctx *)vctx;
    const orc_graph *g = x->g;
    const int64_t *order = x->order;
    const int regularization = x->regularization, learn_non_evidence = x->learn_non_evidence;
    const int64_t truncation = x->truncation;
    int64_t *var_value = x->var_value, *var_value_evid = x->var_value_evid, *G = x->G, *K = x->K, *T = x->T;
    const double *weight_samp = x->weight_samp;
    const uint64_t seed = x->seed, sweep = x->sweep;
    double *Z = (double *)malloc(sizeof(double) * (size_t)max_card(g));
    int rc = ORC_OK;
    {
        for (int64_t i = i0; i < i1 && !rc; i++) {
            int64_t v = order[i];
            const orc_variable *var = &g->variable[v];
            if (var->isEvidence == 4) continue;
            /* learning sweep: one block per variable, counter (q, stream, sweep); stream 0 words 0-1
             * free chain, 2-3 evidence chain; stream 1 words 0-1 the truncation coin */
            uint32_t r[4];
            const uint64_t q = (g->rng_id ? (uint64_t)g->rng_id[v] : (uint64_t)v) & 0xFFFFFFFFFFull;
            orc_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)q, 0u,
                           (uint32_t)sweep, (uint32_t)(sweep >> 32) ^ g->rng_tag, r);
            int64_t evidence, proposal;
            if (var->isEvidence != 1) {
                rc = draw_sample(g, v, Z, var_value_evid, weight_samp, orc_u53(r[2], r[3]), 1,
                                 &evidence);
                if (rc) break;
            } else {
                evidence = var->initialValue;
            }
            var_value_evid[v] = evidence;
            rc = draw_sample(g, v, Z, var_value, weight_samp, orc_u53(r[0], r[1]), 1, &proposal);
            if (rc) break;
            var_value[v] = proposal;
            if (!learn_non_evidence && var->isEvidence != 1) continue;
            int truncate = 0;
            if (regularization == 1) {
                uint32_t t[4];
                orc_philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)q, 1u,
                               (uint32_t)sweep, (uint32_t)(sweep >> 32) ^ g->rng_tag, t);
                truncate = orc_u53(t[0], t[1]) < 1.0 / (double)truncation;
            }
            /* union of the two sorted-unique factor lists, each factor visited once */
            int64_t a, ae, b, be;
            factor_id_range(g, v, evidence, &a, &ae);
            factor_id_range(g, v, proposal, &b, &be);
            if (evidence == proposal || (a == b && ae == be)) { b = be = 0; }
            while ((a < ae || b < be) && !rc) {
                int64_t fa = a < ae ? g->factor_index[a] : INT64_MAX;
                int64_t fb = b < be ? g->factor_index[b] : INT64_MAX;
                int64_t fid = fa < fb ? fa : fb;
                if (fa == fid) a++;
                if (fb == fid) b++;
                rc = accumulate_visit(g, fid, v, evidence, proposal, var_value, var_value_evid,
                                      truncate, G, K, T);
            }
        }
    }
    free(Z);
    return rc;
}

int orc_learn_sweep_dev(const orc_graph *g, const int64_t *order, const int64_t *phase_start,
                        int64_t nphase, double step, int regularization, double reg_param,
                        int64_t truncation, int64_t *var_value, int64_t *var_value_evid,
                        double *weight_value, int learn_non_evidence, uint64_t seed,
                        uint64_t sweep, double cap, double *weight_lag) {
    /* weight_lag != NULL: the device's one-class lag (nsk_set_learn_lag) -- a class samples with the
     * weights as of the end of the class before the previous one (weight_lag), its update moves
     * weight_value on from the previous class's result; the caller sets weight_lag = weight_value at the
     * start of every device call and carries it from sweep to sweep inside one */
    const double *weight_samp = weight_lag ? weight_lag : weight_value;
    int64_t *G = (int64_t *)calloc((size_t)g->nweight, sizeof(int64_t));
    int64_t *K = (int64_t *)calloc((size_t)g->nweight, sizeof(int64_t));
    int64_t *T = (int64_t *)calloc((size_t)g->nweight, sizeof(int64_t));
    int rc = ORC_OK;
    for (int64_t p = 0; p < nphase && !rc; p++) {
        if (phase_start[p] == phase_start[p + 1]) continue;     /* (an empty class is no class: no update, no lag step) */
        {
            orc_learn_ctx ctx = {g, order, regularization, truncation, var_value, var_value_evid, weight_samp,
                                 learn_non_evidence, seed, sweep, G, K, T};
            rc = orc_parallel_range(phase_start[p], phase_start[p + 1], orc_learn_range_dev, &ctx);
        }
        if (rc) break;
        /* end of phase: apply the batch to every touched weight */
        for (int64_t w = 0; w < g->nweight; w++) {
            if (K[w] == 0) {
                if (weight_lag) weight_lag[w] = weight_value[w];
                continue;
            }
            double Gf = (double)G[w] * ldexp(1.0, g->grad_shift - 32);
            double x = weight_value[w];
            if (weight_lag) weight_lag[w] = x;
            /* device-mode step cap: K visits at `step` move the weight by K * step * mean gradient;
             * beyond `cap` the class uses cap / K (DESIGN.md "device-mode learning") */
            double st = step;
            if (cap > 0.0 && (double)K[w] * step > cap) st = cap / (double)K[w];
            if (regularization == 2) {
                double a = 1.0 / (1.0 + reg_param * st);
                x = powi_det(a, (uint64_t)K[w]) * x;
                x = x - st * Gf;
            } else if (regularization == 1) {
                x = x - st * Gf;
                if (T[w] > 0) {
                    double l1 = (reg_param * st * (double)truncation) * (double)T[w];
                    x = (x > 0) ? fmax(0.0, x - l1) : fmin(0.0, x + l1);
                }
            } else {
                x = x - st * Gf;
            }
            weight_value[w] = x;
            G[w] = 0; K[w] = 0; T[w] = 0;
        }
    }
    free(G); free(K); free(T);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * compute_var_map (dataloading.py:16-81)
 * ---------------------------------------------------------------------------------------- */
int orc_compute_var_map(int64_t nvar, orc_variable *variable, int64_t nfactor,
                        const orc_factor *factor, int64_t nedge, const orc_ftv *fmap,
                        int64_t nvtf, orc_vtf *vmap, int64_t nfi, int64_t *factor_index,
                        const uint8_t *domain_mask, const int64_t *factors_to_skip, int64_t nskip) {
    /* 21-30: implicit domains */
    for (int64_t i = 0; i < nvar; i++) {
        if (variable[i].dataType == 0) continue;
        if (domain_mask[i]) continue;
        for (int64_t k = 0; k < variable[i].cardinality; k++)
            vmap[variable[i].vtf_offset + k].value = k;
    }
    /* 34-38: lengths (every edge, skipped factors included, as the reference does) */
    for (int64_t j = 0; j < nedge; j++) {
        int64_t vid = fmap[j].vid;
        int64_t val = variable[vid].dataType == 1 ? fmap[j].dense_equal_to : 0;
        vmap[variable[vid].vtf_offset + val].factor_index_length += 1;
    }
    /* 41-46: offsets */
    int64_t last_len = 0, last_off = 0;
    for (int64_t i = 0; i < nvtf; i++) {
        vmap[i].factor_index_offset = last_off + last_len;
        last_len = vmap[i].factor_index_length;
        last_off = vmap[i].factor_index_offset;
    }
    /* 49-65: scatter */
    int64_t *offsets = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nvtf ? nvtf : 1));
    for (int64_t i = 0; i < nvtf; i++) offsets[i] = vmap[i].factor_index_offset;
    int64_t fts = 0;
    for (int64_t i = 0; i < nfactor; i++) {
        if (fts < nskip && factors_to_skip[fts] == i) { fts++; continue; }
        for (int64_t j = factor[i].ftv_offset; j < factor[i].ftv_offset + factor[i].arity; j++) {
            int64_t vid = fmap[j].vid;
            int64_t val = variable[vid].dataType == 1 ? fmap[j].dense_equal_to : 0;
            int64_t idx = variable[vid].vtf_offset + val;
            if (offsets[idx] >= nfi) { free(offsets); return ORC_E_INDEX; }   /* IndexError */
            factor_index[offsets[idx]] = i;
            offsets[idx] += 1;
        }
    }
    free(offsets);
    /* 68-81: sort + dedupe every slot over its FULL counted length */
    for (int64_t i = 0; i < nvtf; i++) {
        int64_t off = vmap[i].factor_index_offset;
        int64_t len = vmap[i].factor_index_length;
        /* numpy slice semantics: factor_index[off:off+len] is clipped to the array */
        if (off > nfi) off = nfi;
        if (off + len > nfi) len = nfi - off;
        qsort(factor_index + off, (size_t)len, sizeof(int64_t), cmp_i64);
        int64_t n = 0, last = -1;
        for (int64_t k = 0; k < len; k++) {
            int64_t fid = factor_index[off + k];
            if (fid == last) continue;
            last = fid;
            factor_index[off + n++] = fid;
        }
        vmap[i].factor_index_length = n;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline: run_pool semantics (factorgraph.py:13-24) -- T threads, shard formula of
 * inference.py:17-18, shared unsynchronised state (Hogwild), one MT19937 per thread.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const orc_graph *g;
    int tid, nthreads;
    const int64_t *cstart;
    int64_t *count, *var_value, *var_value_evid;
    double *weight_value;
    int sample_evidence, burnin, learn_non_evidence, regularization;
    double step, reg_param;
    int64_t truncation;
    orc_mt np_rng, py_rng;
    double *Z;
    int64_t *fids;
    int rc;
    int learn;
} hog_arg;

static void *hog_worker(void *p) {
    hog_arg *a = (hog_arg *)p;
    if (a->learn)
        a->rc = orc_learn_shard_ref(a->g, a->tid, a->nthreads, a->step, a->regularization,
                                    a->reg_param, a->truncation, a->Z, a->fids, a->var_value,
                                    a->var_value_evid, a->weight_value, a->learn_non_evidence,
                                    &a->np_rng, &a->py_rng);
    else
        a->rc = orc_gibbs_shard_ref(a->g, a->tid, a->nthreads, a->Z, a->cstart, a->count,
                                    a->var_value, a->weight_value, a->sample_evidence, a->burnin,
                                    &a->np_rng);
    return NULL;
}

static int hog_run(hog_arg *args, int nthreads, int64_t nsweeps, double decay) {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    int rc = ORC_OK;
    for (int64_t s = 0; s < nsweeps && !rc; s++) {
        if (nthreads == 1) {
            hog_worker(&args[0]);
        } else {
            for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, hog_worker, &args[t]);
            for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        }
        for (int t = 0; t < nthreads; t++) {
            if (args[t].rc) rc = args[t].rc;
            args[t].step *= decay;
        }
    }
    free(th);
    return rc;
}

static int64_t max_list2(const orc_graph *g) {
    int64_t m = 0;
    for (int64_t i = 0; i < g->nvtf; i++)
        if (g->vmap[i].factor_index_length > m) m = g->vmap[i].factor_index_length;
    return 2 * m + 1;
}

/* per-thread scratch on cache lines of its own: Z is written at every update, and neighbouring
 * 16-byte malloc chunks would make the Hogwild threads share lines */
static void *hog_alloc(size_t bytes) {
    void *p = NULL;
    if (posix_memalign(&p, 128, (bytes + 127) / 128 * 128 + 128)) return NULL;
    return p;
}

int orc_gibbs_hogwild(const orc_graph *g, int nthreads, int64_t nsweeps, const int64_t *cstart,
                      int64_t *count, int64_t *var_value, const double *weight_value,
                      int sample_evidence, int burnin, uint32_t seed) {
    hog_arg *args = (hog_arg *)calloc((size_t)nthreads, sizeof(hog_arg));
    int64_t mc = max_card(g);
    for (int t = 0; t < nthreads; t++) {
        args[t].g = g; args[t].tid = t; args[t].nthreads = nthreads;
        args[t].cstart = cstart; args[t].count = count; args[t].var_value = var_value;
        args[t].weight_value = (double *)weight_value;
        args[t].sample_evidence = sample_evidence; args[t].burnin = burnin;
        args[t].Z = (double *)hog_alloc(sizeof(double) * (size_t)mc);
        orc_mt_seed_numpy(&args[t].np_rng, seed + (uint32_t)t);
    }
    int rc = hog_run(args, nthreads, nsweeps, 1.0);
    for (int t = 0; t < nthreads; t++) free(args[t].Z);
    free(args);
    return rc;
}

int orc_learn_hogwild(const orc_graph *g, int nthreads, int64_t nsweeps, double step, double decay,
                      int regularization, double reg_param, int64_t truncation,
                      int64_t *var_value, int64_t *var_value_evid, double *weight_value,
                      int learn_non_evidence, uint32_t seed) {
    hog_arg *args = (hog_arg *)calloc((size_t)nthreads, sizeof(hog_arg));
    int64_t mc = max_card(g), ml = max_list2(g);
    for (int t = 0; t < nthreads; t++) {
        args[t].g = g; args[t].tid = t; args[t].nthreads = nthreads; args[t].learn = 1;
        args[t].var_value = var_value; args[t].var_value_evid = var_value_evid;
        args[t].weight_value = weight_value; args[t].learn_non_evidence = learn_non_evidence;
        args[t].regularization = regularization; args[t].step = step;
        args[t].reg_param = reg_param; args[t].truncation = truncation;
        args[t].Z = (double *)hog_alloc(sizeof(double) * (size_t)mc);
        args[t].fids = (int64_t *)hog_alloc(sizeof(int64_t) * (size_t)ml);
        orc_mt_seed_numpy(&args[t].np_rng, seed + (uint32_t)t);
        orc_mt_seed_python(&args[t].py_rng, (uint64_t)seed + (uint64_t)t);
    }
    int rc = hog_run(args, nthreads, nsweeps, decay);
    for (int t = 0; t < nthreads; t++) { free(args[t].Z); free(args[t].fids); }
    free(args);
    return rc;
}
