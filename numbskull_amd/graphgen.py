"""Synthetic factor-graph builders and the DeepDive binary writer.

The reference ships one fixture generator, ising/ising.cpp (a C++ program that writes
``graph.{meta,weights,variables,factors}``; format at ising.cpp:88-130, reader at
numbskull/dataloading.py:103-237).  This module produces the same graphs as numpy record
arrays (ready for ``NumbSkull.loadFactorGraph``) with vectorised code that scales to the
10M-50M-variable benchmark configurations, and writes/reads the same on-disk format.

Every builder returns ``(weight, variable, factor, fmap, domain_mask, edges)`` -- the
positional arguments of ``NumbSkull.loadFactorGraph`` (numbskull.py:192-194).
"""

import os

import numpy as np

from .numbskulltypes import Weight, Variable, Factor, FactorToVar

FUNC_EQUAL, FUNC_ISTRUE = 3, 4


def ising_grid(nrows, ncols, weight=0.1, fixed=True, two_weights=False, evidence=None,
               initial=None):
    """N x M Ising grid of binary variables with EQUAL factors to the up and left neighbour.

    Same graph as the (commented-out) generator at ising/ising.cpp:134-199: variable id
    ``i*M+j``; for every cell, in row-major order, first the factor to the cell above
    ``(i*M+j, (i-1)*M+j)`` when ``i>0``, then the factor to the left ``(i*M+j, i*M+j-1)``
    when ``j>0``; featureValue 1, equalPredicate 0.

    ``two_weights``: weight 0 on vertical and weight 1 on horizontal edges (the learning
    variant of benchmark config #3); otherwise a single weight 0.
    ``evidence``: optional int array (nrows*ncols) -> every variable becomes evidence with
    that initial value.  ``initial`` sets initialValue of query variables.
    """
    n, m = int(nrows), int(ncols)
    nvar = n * m
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    if evidence is not None:
        variable["isEvidence"] = 1
        variable["initialValue"] = np.asarray(evidence, np.int64).reshape(nvar)
    elif initial is not None:
        variable["initialValue"] = np.asarray(initial, np.int64).reshape(nvar)

    nw = 2 if two_weights else 1
    wrec = np.zeros(nw, Weight)
    wrec["isFixed"] = bool(fixed)
    wrec["initialValue"] = weight

    # candidate factors per cell in emission order [up, left]; keep the ones that exist
    me = np.arange(nvar, dtype=np.int64)
    exists = np.empty((nvar, 2), np.bool_)
    exists[:, 0] = me >= m                    # i > 0: factor to the cell above
    exists[:, 1] = (me % m) != 0              # j > 0: factor to the left neighbour
    other = np.empty((nvar, 2), np.int64)
    other[:, 0] = me - m
    other[:, 1] = me - 1
    keep = exists.ravel()
    first = np.repeat(me, 2)[keep]
    second = other.ravel()[keep]
    nfactor = int(first.shape[0])

    factor = np.zeros(nfactor, Factor)
    factor["factorFunction"] = FUNC_EQUAL
    factor["featureValue"] = 1.0
    factor["arity"] = 2
    factor["ftv_offset"] = 2 * np.arange(nfactor, dtype=np.int64)
    if two_weights:
        factor["weightId"] = np.tile(np.array([0, 1], np.int64), nvar)[keep]
    vid = np.empty(2 * nfactor, np.int64)
    vid[0::2] = first
    vid[1::2] = second
    fmap = np.zeros(2 * nfactor, FactorToVar)
    fmap["vid"] = vid

    return wrec, variable, factor, fmap, np.zeros(nvar, np.bool_), 2 * nfactor


def ising_pairs(npairs, a=1.0, b=1.0, c=0.5, seed=0):
    """Independent 2-variable evidence pairs for weight recovery (ising/ising.cpp:202-318).

    Pair i is variables (2i, 2i+1), both evidence, drawn from
    p(x,y) ~ exp(a*s(x) + b*s(y) + c*s(x==y)), s(t)=+1 if t else -1; three free weights
    (initial 0): ISTRUE(x) -> w0, ISTRUE(y) -> w1, EQUAL(x,y) -> w2.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    z = np.exp([-a - b + c, -a + b - c, a - b - c, a + b + c])
    idx = np.searchsorted(np.cumsum(z) / z.sum(), rng.random(npairs), side="left").clip(0, 3)
    nvar = 2 * npairs
    variable = np.zeros(nvar, Variable)
    variable["isEvidence"] = 1
    variable["cardinality"] = 2
    variable["initialValue"][0::2] = (idx >= 2)
    variable["initialValue"][1::2] = (idx % 2 == 1)
    wrec = np.zeros(3, Weight)
    factor = np.zeros(3 * npairs, Factor)
    factor["featureValue"] = 1.0
    factor["factorFunction"] = np.tile([FUNC_ISTRUE, FUNC_ISTRUE, FUNC_EQUAL], npairs)
    factor["weightId"] = np.tile([0, 1, 2], npairs)
    factor["arity"] = np.tile([1, 1, 2], npairs)
    factor["ftv_offset"] = np.cumsum(factor["arity"]) - factor["arity"]
    fmap = np.zeros(4 * npairs, FactorToVar)
    base = 2 * np.arange(npairs, dtype=np.int64)
    fmap["vid"][0::4] = base
    fmap["vid"][1::4] = base + 1
    fmap["vid"][2::4] = base
    fmap["vid"][3::4] = base + 1
    return wrec, variable, factor, fmap, np.zeros(nvar, np.bool_), 4 * npairs


def lf_graph(prior, accuracy, copies, seed=0):
    """Data-programming generative model used by the reference's test_lf_learning.py:22-126.

    ``copies`` x (one hidden binary label y + n labelling-function outputs, cardinality 3,
    dataType 0, evidence).  Factors: DP_GEN_CLASS_PRIOR(y) -> weight 0 (initial 0) and
    DP_GEN_LF_ACCURACY(y, lf_i) -> weight i+1 (initial 1).  Evidence is sampled from the
    exact joint with ``prior``/``accuracy`` planted.
    """
    n = len(accuracy)
    rng = np.random.Generator(np.random.PCG64(seed))
    states = 2 * 3 ** n
    logp = np.zeros(states)
    decoded = np.zeros((states, n + 1), np.int64)
    for s in range(states):
        y, rest = s % 2, s // 2
        decoded[s, 0] = y
        e = prior * (2 * y - 1)
        for j in range(n):
            lf = rest % 3
            rest //= 3
            decoded[s, j + 1] = lf
            e += accuracy[j] * (lf - 1) * (2 * y - 1)
        logp[s] = e
    cdf = np.cumsum(np.exp(logp))
    cdf /= cdf[-1]
    pick = np.searchsorted(cdf, rng.random(copies), side="left").clip(0, states - 1)

    nvar = copies * (1 + n)
    nfac = copies * (1 + n)
    nedge = copies * (1 + 2 * n)
    wrec = np.zeros(1 + n, Weight)
    wrec["initialValue"] = 1.0
    wrec["initialValue"][0] = 0.0
    variable = np.zeros(nvar, Variable)
    factor = np.zeros(nfac, Factor)
    fmap = np.zeros(nedge, FactorToVar)
    factor["featureValue"] = 1.0
    for cp in range(copies):
        v0, f0, e0 = cp * (1 + n), cp * (1 + n), cp * (1 + 2 * n)
        variable[v0]["cardinality"] = 2
        factor[f0]["factorFunction"] = 18
        factor[f0]["arity"] = 1
        factor[f0]["ftv_offset"] = e0
        fmap[e0]["vid"] = v0
        for i in range(n):
            vi = v0 + 1 + i
            variable[vi]["isEvidence"] = 1
            variable[vi]["initialValue"] = decoded[pick[cp], 1 + i]
            variable[vi]["cardinality"] = 3
            fi = f0 + 1 + i
            factor[fi]["factorFunction"] = 21
            factor[fi]["weightId"] = 1 + i
            factor[fi]["arity"] = 2
            factor[fi]["ftv_offset"] = e0 + 1 + 2 * i
            fmap[e0 + 1 + 2 * i]["vid"] = v0
            fmap[e0 + 2 + 2 * i]["vid"] = vi
    return wrec, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


LR_BLOCK = 65536     # ids per generator block of mixed_lr_graph: every block draws from streams of its own


def _block_map(fn, items):
    """``[fn(x) for x in items]`` over a pool of host threads (numpy's generators and array kernels release the
    GIL): every id block of the LR generator draws from streams of its own, so the blocks are independent and
    the result does not depend on the thread count.  NSK_GEN_THREADS (default: the hardware's, at most 32)."""
    items = list(items)
    import os
    nthreads = int(os.environ.get("NSK_GEN_THREADS", "0")) or min(32, os.cpu_count() or 1)
    if nthreads <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(nthreads, len(items))) as pool:
        return list(pool.map(fn, items))


def _lr_variables(nvar, seed, cat_frac, evidence_frac, block):
    """Compact per-variable attributes of the WHOLE graph (4 bytes per variable): every id block has a
    stream of its own, so any rank reproduces them without the factors."""
    card = np.empty(nvar, np.uint8)
    ev = np.empty(nvar, np.bool_)
    init = np.empty(nvar, np.uint8)
    nf_of = np.empty(nvar, np.uint8)

    def one(bv):
        b, v0 = bv
        n = min(block, nvar - v0)
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), 0, b])))
        is_cat = rng.random(n) < cat_frac
        c = np.where(is_cat, rng.integers(3, 9, n), 2)
        e = rng.random(n) < evidence_frac
        card[v0:v0 + n] = c
        ev[v0:v0 + n] = e
        init[v0:v0 + n] = np.where(e, (rng.random(n) * c).astype(np.int64), 0)
        nf_of[v0:v0 + n] = 1 + np.minimum(rng.poisson(2.0, n), 15)
    _block_map(one, enumerate(range(0, nvar, block)))
    return card, ev, init, nf_of


def _lr_factor_block(seed, b, v0, nf_of_b, card, nvar, nweights, window, global_frac):
    """Factors headed by the variables of id block ``b`` (global member ids), from the block's own stream."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), 1, b])))
    n = len(nf_of_b)
    nfactor = int(nf_of_b.sum(dtype=np.int64))
    head = np.repeat(np.arange(v0, v0 + n, dtype=np.int64), nf_of_b)
    arity = rng.choice(np.array([1, 2, 3, 4]), size=nfactor, p=[.4, .3, .2, .1]).astype(np.int64)
    off = np.cumsum(arity) - arity
    nedge = int(arity.sum())
    fac_of_edge = np.repeat(np.arange(nfactor, dtype=np.int64), arity)
    pos = np.arange(nedge, dtype=np.int64) - off[fac_of_edge]
    is_head = pos == arity[fac_of_edge] - 1
    h = head[fac_of_edge]
    local = np.clip(h + rng.integers(-window, window + 1, nedge), 0, nvar - 1)
    glob = rng.integers(0, nvar, nedge)
    other = np.where(rng.random(nedge) < global_frac, glob, local)
    vid = np.where(is_head, h, other)
    cv = card[vid].astype(np.int64)
    deo = (rng.random(nedge) * cv).astype(np.int64)
    any_cat = np.add.reduceat((cv > 2).astype(np.int32), off) > 0
    r = rng.random(nfactor)
    func = np.where(arity == 1, FUNC_ISTRUE, np.where(r < 0.5, 1, 13))          # OR / IMPLY_MLN
    func_cat = np.where(r < 1 / 3, 14, np.where(r < 2 / 3, 17, 12))             # OR_CAT/IMPLY_MLN_CAT/AND_CAT
    func = np.where(any_cat, func_cat, func)
    wid = np.minimum((nweights * rng.random(nfactor) ** 2).astype(np.int64), nweights - 1)
    return func, wid, arity, off, vid, deo


def _lr_records(func, wid, arity, vid, deo):
    """The packed reference records (numbskulltypes.py:11-39) from the columns; filled chunk by chunk over the
    host threads (strided stores into 34- and 16-byte records: the largest serial lap of the generator)."""
    nf, ne = len(func), len(vid)
    factor = np.empty(nf, Factor)
    fmap = np.empty(ne, FactorToVar)
    ftv = np.cumsum(arity) - arity
    step = 1 << 20

    def fill_f(a):
        b = min(nf, a + step)
        f = factor[a:b]
        f["factorFunction"] = func[a:b]
        f["weightId"] = wid[a:b]
        f["featureValue"] = 1.0
        f["arity"] = arity[a:b]
        f["ftv_offset"] = ftv[a:b]

    def fill_e(a):
        b = min(ne, a + step)
        m = fmap[a:b]
        m["vid"] = vid[a:b]
        m["dense_equal_to"] = deo[a:b]
    _block_map(lambda t: (fill_f if t[0] == 0 else fill_e)(t[1]),
               [(0, a) for a in range(0, nf, step)] + [(1, a) for a in range(0, ne, step)])
    return factor, fmap


def _lr_nweights(nvar, nweights):
    return max(1, min(10 ** 6, nvar // 50)) if nweights is None else int(nweights)


def mixed_lr_graph(nvar, seed=20240603, nweights=None, window=1024, global_frac=0.01,
                   evidence_frac=0.5, cat_frac=0.25, block=LR_BLOCK):
    """Mixed-arity logistic-regression-style graph (benchmark config #5).

    75 % boolean / 25 % categorical (dataType 1, cardinality 3..8) variables, half of them
    evidence.  Each variable v heads ``1+min(Poisson(2),15)`` factors of arity 1..4
    (p = .4/.3/.2/.1) whose other members are drawn from the id window ``[v-1024, v+1024]``
    (99 %) or uniformly (1 %).  Function: arity 1 -> ISTRUE; all-boolean -> OR or IMPLY_MLN;
    any categorical member -> OR_CAT / IMPLY_MLN_CAT / AND_CAT, with ``dense_equal_to``
    uniform over each member's domain.  ``weightId = floor(nweights*u^2)``, all free,
    initial 0.  IMPLY_MLN* need the library's ``head_by_vid`` lookup (the reference's
    literal head indexing is out of range on such graphs, inference.py:243).

    The generator is keyed per id block (``block`` ids: PCG64 streams seeded (seed, 0, b) for the
    block's variables and (seed, 1, b) for the factors they head), so a rank of an N-rank run can
    produce its shard alone (``mixed_lr_shard``) -- the reference's minions load only their partition
    (salt/src/numbskull_minion.py:185).
    """
    nvar = int(nvar)
    nweights = _lr_nweights(nvar, nweights)
    card, ev, init, nf_of = _lr_variables(nvar, seed, cat_frac, evidence_frac, block)
    variable = np.zeros(nvar, Variable)
    variable["dataType"] = card > 2
    variable["cardinality"] = card
    variable["isEvidence"] = ev
    variable["initialValue"] = init
    parts = _block_map(lambda bv: _lr_factor_block(seed, bv[0], bv[1], nf_of[bv[1]:bv[1] + block], card, nvar, nweights,
                                                   window, global_frac), enumerate(range(0, nvar, block)))
    cat = lambda i: np.concatenate([p_[i] for p_ in parts]) if parts else np.zeros(0, np.int64)
    cols = _block_map(cat, (0, 1, 2, 4, 5))
    del parts
    factor, fmap = _lr_records(*cols)
    wrec = np.zeros(nweights, Weight)
    return wrec, variable, factor, fmap, np.zeros(nvar, np.bool_), len(fmap)


def _lr_shard_from(kept, card, ev, init, lo, hi, nweights):
    """The shard's arrays from the kept (func, wid, arity, vid, deo) pieces of the factor blocks."""
    cat = lambda i: np.concatenate([k[i] for k in kept]) if kept else np.zeros(0, np.int64)
    vid = cat(3)
    gids = np.unique(np.concatenate([np.arange(lo, hi, dtype=np.int64), vid]))
    lvar = np.zeros(len(gids), Variable)
    lvar["dataType"] = card[gids] > 2
    lvar["cardinality"] = card[gids]
    lvar["isEvidence"] = ev[gids]
    lvar["initialValue"] = init[gids]
    lvar["isEvidence"][(gids < lo) | (gids >= hi)] = 4
    factor, fmap = _lr_records(cat(0), cat(1), cat(2), np.searchsorted(gids, vid), cat(4))
    l0, l1 = int(np.searchsorted(gids, lo)), int(np.searchsorted(gids, hi))
    return (np.zeros(nweights, Weight), lvar, factor, fmap, np.zeros(len(gids), np.bool_), len(fmap)), gids, (l0, l1)


def mixed_lr_shards(nvar, ranges, seed=20240603, nweights=None, window=1024, global_frac=0.01,
                    evidence_frac=0.5, cat_frac=0.25, block=LR_BLOCK):
    """``[mixed_lr_shard(nvar, lo, hi, ...) for lo, hi in ranges]`` in ONE pass over the factor blocks: for a
    process that holds several shards at once (the 8-handles-on-one-device tests).  A rank of a real run calls
    ``mixed_lr_shard`` for its own range."""
    nvar = int(nvar)
    ranges = [(int(lo), int(hi)) for lo, hi in ranges]
    nweights = _lr_nweights(nvar, nweights)
    card, ev, init, nf_of = _lr_variables(nvar, seed, cat_frac, evidence_frac, block)
    def one(bv):
        b, v0 = bv
        func, wid, arity, off, vid, deo = _lr_factor_block(seed, b, v0, nf_of[v0:v0 + block], card, nvar,
                                                           nweights, window, global_frac)
        out = []
        for lo, hi in ranges:
            keep_f = np.add.reduceat(((vid >= lo) & (vid < hi)).astype(np.int32), off) > 0
            if not keep_f.any():
                out.append(None)
                continue
            keep_e = np.repeat(keep_f, arity)
            out.append((func[keep_f], wid[keep_f], arity[keep_f], vid[keep_e], deo[keep_e]))
        return out
    per_block = _block_map(one, enumerate(range(0, nvar, block)))
    kept = [[pb[r] for pb in per_block if pb[r] is not None] for r in range(len(ranges))]
    del per_block
    return [_lr_shard_from(kept[r], card, ev, init, lo, hi, nweights) for r, (lo, hi) in enumerate(ranges)]


def mixed_lr_shard(nvar, lo, hi, seed=20240603, nweights=None, window=1024, global_frac=0.01,
                   evidence_frac=0.5, cat_frac=0.25, block=LR_BLOCK):
    """``extract_shard(mixed_lr_graph(nvar, ...), lo, hi)`` without ever holding the whole graph: the
    compact attributes of all variables (4 bytes each), then the factor blocks one at a time, keeping
    the factors with a member in ``[lo, hi)``.  Same return value as ``extract_shard``."""
    nvar, lo, hi = int(nvar), int(lo), int(hi)
    nweights = _lr_nweights(nvar, nweights)
    card, ev, init, nf_of = _lr_variables(nvar, seed, cat_frac, evidence_frac, block)
    def one(bv):
        b, v0 = bv
        func, wid, arity, off, vid, deo = _lr_factor_block(seed, b, v0, nf_of[v0:v0 + block], card, nvar,
                                                           nweights, window, global_frac)
        keep_f = np.add.reduceat(((vid >= lo) & (vid < hi)).astype(np.int32), off) > 0
        if not keep_f.any():
            return None
        keep_e = np.repeat(keep_f, arity)
        return func[keep_f], wid[keep_f], arity[keep_f], vid[keep_e], deo[keep_e]
    kept = [k for k in _block_map(one, enumerate(range(0, nvar, block))) if k is not None]
    return _lr_shard_from(kept, card, ev, init, lo, hi, nweights)


def boolean_weighted_graph(nvar, seed=0, window=64, factors_per_var=2.0, max_arity=3):
    """Boolean graph with one weight per factor: for every variable an ISTRUE prior, plus
    ``factors_per_var * nvar`` OR / EQUAL factors of arity 2..``max_arity`` over nearby variables (ids within
    ``window``).  The shape of feature-weighted DeepDive graphs whose factors carry individual
    weights; all weights fixed at small random values so that inference is well-conditioned."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nvar = int(nvar)
    nextra = int(factors_per_var * nvar)
    arity = np.concatenate([np.ones(nvar, np.int64), rng.integers(2, max_arity + 1, nextra)])
    nfactor = len(arity)
    off = np.cumsum(arity) - arity
    nedge = int(arity.sum())
    func = np.concatenate([np.full(nvar, FUNC_ISTRUE, np.int64),
                           np.where(rng.random(nextra) < 0.5, 1, FUNC_EQUAL)])
    fac_of_edge = np.repeat(np.arange(nfactor, dtype=np.int64), arity)
    pos = np.arange(nedge, dtype=np.int64) - off[fac_of_edge]
    anchor = np.concatenate([np.arange(nvar, dtype=np.int64), rng.integers(0, nvar, nextra)])
    # distinct members inside a factor: anchor, anchor+d1, anchor+d1+d2 (mod nvar)
    step = rng.integers(1, window, nedge)
    step[pos == 0] = 0
    cum = np.cumsum(step)
    vid = (anchor[fac_of_edge] + cum - cum[off[fac_of_edge]]) % nvar
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    factor = np.zeros(nfactor, Factor)
    factor["factorFunction"] = func
    factor["weightId"] = np.arange(nfactor)
    factor["featureValue"] = 1.0
    factor["arity"] = arity
    factor["ftv_offset"] = off
    fmap = np.zeros(nedge, FactorToVar)
    fmap["vid"] = vid
    wrec = np.zeros(nfactor, Weight)
    wrec["isFixed"] = True
    wrec["initialValue"] = rng.normal(0, 0.3, nfactor)
    return wrec, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


# --------------------------------------------------------------------------------------------
# DeepDive binary format (big-endian), SURVEY.md Appendix B
# --------------------------------------------------------------------------------------------
_W_DISK = np.dtype([("weightId", ">i8"), ("isFixed", "u1"), ("initialValue", ">f8")])
_V_DISK = np.dtype([("variableId", ">i8"), ("isEvidence", "u1"), ("initialValue", ">i8"),
                    ("dataType", ">i2"), ("cardinality", ">i8")])
assert _W_DISK.itemsize == 17 and _V_DISK.itemsize == 27


def ising_grid_shard(nrows, ncols, lo, hi, weight=0.1, fixed=True, two_weights=False, evidence=None):
    """``extract_shard(ising_grid(nrows, ncols, ...), lo, hi)`` without the whole grid: the cells ``[lo, hi + ncols)`` are
    the only ones whose up / left factors can touch an owned variable, so a rank of an N-rank run builds its shard from
    O(shard) memory (the 100M grid on 8 ranks: 12.5M cells each instead of 8 x 100M).  ``evidence``: None, or a callable
    ``global ids -> values`` (every variable evidence, config #3's learning variant)."""
    n, m = int(nrows), int(ncols)
    nvar = n * m
    lo, hi = int(lo), int(hi)
    me = np.arange(lo, min(nvar, hi + m), dtype=np.int64)               # candidate cells, row-major like ising_grid
    exists = np.empty((len(me), 2), np.bool_)
    exists[:, 0] = me >= m
    exists[:, 1] = (me % m) != 0
    other = np.empty((len(me), 2), np.int64)
    other[:, 0] = me - m
    other[:, 1] = me - 1
    first = np.repeat(me, 2)
    second = other.ravel()
    wid = np.tile(np.array([0, 1], np.int64), len(me))
    owned = lambda x: (x >= lo) & (x < hi)
    keep = exists.ravel() & (owned(first) | owned(second))
    first, second, wid = first[keep], second[keep], wid[keep]
    nfactor = int(first.shape[0])
    gids = np.unique(np.concatenate([np.arange(lo, hi, dtype=np.int64), first, second]))
    variable = np.zeros(len(gids), Variable)
    variable["cardinality"] = 2
    if evidence is not None:
        variable["isEvidence"] = 1
        variable["initialValue"] = np.asarray(evidence(gids), np.int64)
    variable["isEvidence"][(gids < lo) | (gids >= hi)] = 4
    wrec = np.zeros(2 if two_weights else 1, Weight)
    wrec["isFixed"] = bool(fixed)
    wrec["initialValue"] = weight
    factor = np.zeros(nfactor, Factor)
    factor["factorFunction"] = FUNC_EQUAL
    factor["featureValue"] = 1.0
    factor["arity"] = 2
    factor["ftv_offset"] = 2 * np.arange(nfactor, dtype=np.int64)
    if two_weights:
        factor["weightId"] = wid
    vid = np.empty(2 * nfactor, np.int64)
    vid[0::2] = first
    vid[1::2] = second
    fmap = np.zeros(2 * nfactor, FactorToVar)
    fmap["vid"] = np.searchsorted(gids, vid)
    l0, l1 = int(np.searchsorted(gids, lo)), int(np.searchsorted(gids, hi))
    return (wrec, variable, factor, fmap, np.zeros(len(gids), np.bool_), 2 * nfactor), gids, (l0, l1)


def replicate(g, initial_values):
    """R disjoint copies of the graph `g` that share its weights; copy r takes its variables' initialValue from
    initial_values[r] (evidence configurations of a small model: the learning tests)."""
    w, v, f, fm, dm, edges = g
    R, nv, nf, ne = len(initial_values), len(v), len(f), len(fm)
    V = np.tile(v, R)
    V["initialValue"] = np.concatenate([np.asarray(x, np.int64) for x in initial_values])
    F = np.tile(f, R)
    F["ftv_offset"] += np.repeat(np.arange(R, dtype=np.int64) * ne, nf)
    FM = np.tile(fm, R)
    FM["vid"] += np.repeat(np.arange(R, dtype=np.int64) * nv, ne)
    return w.copy(), V, F, FM, np.tile(dm, R), R * int(edges)


def extract_shard(g, lo, hi):
    """The part of a graph the shard that owns variables ``[lo, hi)`` needs -- what the reference's
    minions load (salt/src/numbskull_minion.py:185): the owned variables, the variables outside the
    range their factors touch (kept as ghosts, ``isEvidence = 4``: read, never sampled,
    inference.py:21-23) and every factor with an owned member, all renumbered locally in ascending
    global order (so factor lists keep their order and a shard samples exactly what it samples when
    it is handed the whole graph with ``own_range``).  Weights keep their ids.

    Returns ``(weight, variable, factor, fmap, domain_mask, edges), global_ids, (l0, l1)``:
    ``global_ids[i]`` = the global id of local variable ``i``; ``[l0, l1)`` = the owned variables'
    local ids.  Factors whose head is looked up at its literal edge index (IMPLY_MLN & co. without
    ``head_by_vid``, inference.py:243) do not survive renumbering: use ``head_by_vid``.
    """
    weight, variable, factor, fmap, domain_mask, edges = g
    vid = fmap["vid"]
    arity = factor["arity"].astype(np.int64)
    nfactor = len(factor)
    fac_of_edge = np.repeat(np.arange(nfactor, dtype=np.int64), arity)
    owned_edge = (vid >= lo) & (vid < hi)
    keep_f = np.zeros(nfactor, np.bool_)
    keep_f[fac_of_edge[owned_edge]] = True
    keep_e = keep_f[fac_of_edge]
    gids = np.unique(np.concatenate([np.arange(lo, hi, dtype=np.int64), vid[keep_e]]))
    lvar = variable[gids].copy()
    ghost = (gids < lo) | (gids >= hi)
    lvar["isEvidence"][ghost] = 4
    lfac = factor[keep_f].copy()
    lar = lfac["arity"].astype(np.int64)
    lfac["ftv_offset"] = np.cumsum(lar) - lar
    lfm = fmap[keep_e].copy()
    lfm["vid"] = np.searchsorted(gids, lfm["vid"])
    l0, l1 = int(np.searchsorted(gids, lo)), int(np.searchsorted(gids, hi))
    return (weight, lvar, lfac, lfm, np.ascontiguousarray(domain_mask[gids]), int(lar.sum())), gids, (l0, l1)


def write_graph(directory, weight, variable, factor, fmap, domains=None):
    """Write ``graph.{meta,weights,variables,factors[,domains]}`` in the format the
    reference reads (dataloading.py:103-237) and ising/ising.cpp:88-130 writes.

    ``domains``: optional ``{vid: sorted int array}``; when given, ``fmap.dense_equal_to``
    and ``variable.initialValue`` are taken to be DENSE indices and are written back as
    the original domain values (the loader re-maps them, dataloading.py:162-185,213-218).
    """
    os.makedirs(directory, exist_ok=True)
    nedge = int(factor["arity"].sum())
    with open(os.path.join(directory, "graph.meta"), "w") as f:
        f.write("%d,%d,%d,%d" % (len(weight), len(variable), len(factor), nedge))

    w = np.zeros(len(weight), _W_DISK)
    w["weightId"] = np.arange(len(weight))
    w["isFixed"] = weight["isFixed"]
    w["initialValue"] = weight["initialValue"]
    w.tofile(os.path.join(directory, "graph.weights"))

    v = np.zeros(len(variable), _V_DISK)
    v["variableId"] = np.arange(len(variable))
    v["isEvidence"] = variable["isEvidence"].astype(np.uint8)
    init = variable["initialValue"].copy()
    if domains:
        for vid, dom in domains.items():
            init[vid] = np.asarray(dom, np.int64)[init[vid]]
    v["initialValue"] = init
    v["dataType"] = variable["dataType"]
    v["cardinality"] = variable["cardinality"]
    v.tofile(os.path.join(directory, "graph.variables"))

    # graph.factors: i2 func, i8 arity, arity x (i8 vid, i8 value), i8 weightId, f8 featureValue
    arity = factor["arity"].astype(np.int64)
    rec_len = 2 + 8 + 16 * arity + 16
    start = np.cumsum(rec_len) - rec_len
    buf = np.zeros(int(rec_len.sum()), np.uint8)

    def put(offsets, values, dt):
        raw = np.ascontiguousarray(values.astype(dt)).view(np.uint8).reshape(len(values), -1)
        idx = offsets[:, None] + np.arange(raw.shape[1])[None, :]
        buf[idx] = raw

    put(start, factor["factorFunction"], ">i2")
    put(start + 2, arity, ">i8")
    fac_of_edge = np.repeat(np.arange(len(factor), dtype=np.int64), arity)
    pos = np.arange(nedge, dtype=np.int64) - (np.cumsum(arity) - arity)[fac_of_edge]
    src = factor["ftv_offset"][fac_of_edge] + pos
    eoff = start[fac_of_edge] + 10 + 16 * pos
    vids = fmap["vid"][src]
    vals = fmap["dense_equal_to"][src].copy()
    if domains:
        for vid, dom in domains.items():
            sel = vids == vid
            vals[sel] = np.asarray(dom, np.int64)[vals[sel]]
    put(eoff, vids, ">i8")
    put(eoff + 8, vals, ">i8")
    tail = start + 10 + 16 * arity
    put(tail, factor["weightId"], ">i8")
    put(tail + 8, factor["featureValue"], ">f8")
    buf.tofile(os.path.join(directory, "graph.factors"))

    if domains:
        with open(os.path.join(directory, "graph.domains"), "wb") as f:
            for vid in sorted(domains):
                dom = np.asarray(domains[vid], np.int64)
                np.array([vid, len(dom)], ">i8").tofile(f)
                dom.astype(">i8").tofile(f)


PF_FUNCS = {1: 0, 2: 1, 4: 1}      # OR -> "some member is 1"; AND, ISTRUE -> "no member is 0" (messages.py:1335-1337)


def partial_factors(shard, gids, own, nvar_global, world):
    """Partial factors for one shard of a range partition (the reference's PF surgery,
    salt/src/messages.py:1083-1206, and its per-epoch values, :1333-1355).

    ``shard, gids, own`` as returned by ``extract_shard`` / ``mixed_lr_shard``.  Every factor OR / AND / ISTRUE of the
    shard that has TWO OR MORE members owned by one foreign shard q gets them replaced by ONE boolean ghost variable
    (isEvidence 4) that stands for their aggregate -- "some member is 1" for OR, "no member is 0" for AND / ISTRUE --,
    which q computes after every sweep and ships instead of the members' values (``nsk_pf_setup``).  The factor's
    value is exactly what it is with the members themselves: OR(a.., b..) = OR(a.., OR(b..)), and AND likewise.
    Factors with the same function class over the same foreign members share one aggregate.

    Returns ``(shard', gids', own, pf)``: the rewritten shard; ``gids'`` = ``gids`` followed by synthetic global ids
    ``nvar_global + k`` for the aggregates (they sort behind every real variable, like their local ids); ``pf`` = list of
    ``(owner q, op, member global ids, local id of the ghost)`` in local-id order.  Ghost variables whose every use went
    into an aggregate stay in the arrays unread (the library exchanges only what is read)."""
    weight, variable, factor, fmap, domain_mask, edges = shard
    gids = np.asarray(gids, np.int64)
    l0, l1 = own
    n = int(nvar_global)
    bounds = (np.arange(world + 1, dtype=np.int64) * n) // world
    owner_of = lambda g: int(np.searchsorted(bounds, g, side="right") - 1)
    me = owner_of(int(gids[l0])) if l1 > l0 else -1
    vid = fmap["vid"].astype(np.int64)
    arity = factor["arity"].astype(np.int64)
    off = factor["ftv_offset"].astype(np.int64)
    fac_of_edge = np.repeat(np.arange(len(factor), dtype=np.int64), arity)
    foreign_edge = (vid < l0) | (vid >= l1)
    nforeign = np.bincount(fac_of_edge[foreign_edge], minlength=len(factor))
    func = factor["factorFunction"].astype(np.int64)
    cand = np.nonzero((nforeign >= 2) & np.isin(func, list(PF_FUNCS)))[0]
    groups = {}                                 # (q, op, members) -> [(factor, member positions)]
    for f in cand.tolist():
        mem = vid[off[f]:off[f] + arity[f]]
        by_owner = {}
        for pos, m in enumerate(mem.tolist()):
            if l0 <= m < l1:
                continue
            by_owner.setdefault(owner_of(int(gids[m])), []).append(pos)
        for q, poss in by_owner.items():
            if len(poss) < 2:
                continue
            key = (q, PF_FUNCS[int(func[f])], tuple(sorted(set(int(gids[mem[p]]) for p in poss))))
            groups.setdefault(key, []).append((f, poss))
    if not groups:
        return shard, gids, own, []
    keys = sorted(groups)
    nloc = len(variable)
    lvar = np.concatenate([variable, np.zeros(len(keys), variable.dtype)])
    local_of = {}
    pf = []
    pos_of_gid = lambda g: int(np.searchsorted(gids[:nloc], g))
    for k, key in enumerate(keys):
        q, op, members = key
        lid = nloc + k
        local_of[key] = lid
        inits = [int(variable["initialValue"][pos_of_gid(g)]) for g in members]
        lvar[lid]["isEvidence"] = 4
        lvar[lid]["dataType"] = 0
        lvar[lid]["cardinality"] = 2
        lvar[lid]["initialValue"] = int(any(x == 1 for x in inits)) if op == 0 else int(all(x != 0 for x in inits))
        pf.append((q, op, np.asarray(members, np.int64), lid))
    # rewrite the member lists of the factors concerned
    drop = np.zeros(len(vid), np.bool_)
    new_vid = vid.copy()
    for key, uses in groups.items():
        for f, poss in uses:
            new_vid[off[f] + poss[0]] = local_of[key]          # the aggregate takes the first member's place
            for p in poss[1:]:
                drop[off[f] + p] = True
    keep = ~drop
    lfm = fmap[keep].copy()
    lfm["vid"] = new_vid[keep]
    lfm["dense_equal_to"][lfm["vid"] >= nloc] = 0
    lfac = factor.copy()
    lar = arity - np.bincount(fac_of_edge[drop], minlength=len(factor))
    lfac["arity"] = lar
    lfac["ftv_offset"] = np.cumsum(lar) - lar
    dm = np.concatenate([np.asarray(domain_mask), np.zeros(len(keys), np.asarray(domain_mask).dtype)])
    gids2 = np.concatenate([gids, n + np.arange(len(keys), dtype=np.int64)])
    return (weight, lvar, lfac, lfm, dm, int(lar.sum())), gids2, own, pf


def voter_graph(nheads, width=12, seed=0, nweights=8):
    """"Voter" graph (the shape of the reference's experiments/ generators and of the graphs partial factors were made
    for): ``nheads`` head variables, ids ``[0, nheads)``, each under ONE clause -- OR or AND -- over
    ``width - 1`` voter variables of its own (ids ``nheads + i (width - 1) ...``, used by no other clause) and itself as
    the last member; every voter also has an ISTRUE prior.  ``nweights`` shared free weights, half the heads evidence.
    Range-partitioned, the shards that own the heads read every voter across the cut -- or one aggregate per clause."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nheads, k = int(nheads), int(width) - 1
    nvar = nheads * (k + 1)
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    variable["isEvidence"][:nheads] = rng.random(nheads) < 0.5
    variable["initialValue"] = rng.integers(0, 2, nvar)
    heads = np.arange(nheads, dtype=np.int64)
    voters = nheads + heads[:, None] * k + np.arange(k, dtype=np.int64)[None, :]
    clause_vid = np.concatenate([voters, heads[:, None]], axis=1).reshape(-1)
    prior = np.arange(nheads, nvar, dtype=np.int64)
    nf = nheads + len(prior)
    factor = np.zeros(nf, Factor)
    factor["factorFunction"][:nheads] = np.where(rng.random(nheads) < 0.5, 1, 2)        # OR / AND
    factor["factorFunction"][nheads:] = FUNC_ISTRUE
    factor["arity"][:nheads] = k + 1
    factor["arity"][nheads:] = 1
    factor["weightId"] = rng.integers(0, nweights, nf)
    factor["featureValue"] = 1.0
    ar = factor["arity"].astype(np.int64)
    factor["ftv_offset"] = np.cumsum(ar) - ar
    fmap = np.zeros(int(ar.sum()), FactorToVar)
    fmap["vid"] = np.concatenate([clause_vid, prior])
    weight = np.zeros(nweights, Weight)
    weight["initialValue"] = rng.normal(0, 0.3, nweights)
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), len(fmap)
