#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (HazyResearch/numbskull, mounted read-only at /root/reference) is imported in
its sanctioned pure-Python mode -- its own CI runs every test with NUMBA_DISABLE_JIT=1
(.travis.yml:55-57) -- through the identity ``numba.jit`` stand-in under
tools/oracle_shim/.  Nothing from the reference is copied: the fixtures hold only inputs
(record arrays built by numbskull_amd.graphgen / this script) and the outputs the
reference computed for them.  The one exception is tests/golden/test_coin/, the 1.4 kB
binary data files of the reference's own ``test/`` fixture (data, not source).

Seeding: the reference has no seed parameter; in pure-Python mode its draws come from
numpy's global legacy MT19937 (np.random.rand(), inference.py:50) and Python's ``random``
(learning.py:90), so ``np.random.seed(s); random.seed(s)`` pins a run.

Usage:  python tools/make_goldens.py        (exits 0 with a note if /root/reference is absent)
"""

import io
import os
import random
import shutil
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("NSK_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

if not os.path.isdir(os.path.join(REF, "numbskull")):
    print("reference tree not present at %s: nothing to do" % REF)
    sys.exit(0)

sys.path.insert(0, os.path.join(HERE, "oracle_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numbskull as ref                                   # noqa: E402  (the reference)
from numbskull import inference as ref_inf                # noqa: E402
from numbskull import dataloading as ref_dl               # noqa: E402
from numbskull.numbskulltypes import (Weight, Variable, Factor, FactorToVar,     # noqa: E402
                                      VarToFactor)
from numbskull_amd import graphgen                        # noqa: E402  (our builders)


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def seed_all(s):
    np.random.seed(s)
    random.seed(s)


def graph_arrays(prefix, g):
    w, v, f, fm, dm, edges = g
    return {prefix + "weight": w.copy(), prefix + "variable": v.copy(), prefix + "factor": f.copy(),
            prefix + "fmap": fm.copy(), prefix + "domain_mask": dm.copy(),
            prefix + "edges": np.int64(edges)}


def load_ref(g, factors_to_skip=None, **kw):
    ns = ref.NumbSkull(quiet=True, **kw)
    w, v, f, fm, dm, edges = [x.copy() if isinstance(x, np.ndarray) else x for x in g]
    if factors_to_skip is None:
        quiet(ns.loadFactorGraph, w, v, f, fm, dm, int(edges))
    else:
        quiet(ns.loadFactorGraph, w, v, f, fm, dm, int(edges),
              factors_to_skip=np.asarray(factors_to_skip, np.int64))
    return ns, ns.factorGraphs[0]


# --------------------------------------------------------------------------------------------
# G1  eval_factor truth tables (pins SURVEY row a4)
# --------------------------------------------------------------------------------------------
def g1_eval_factor():
    rng = np.random.Generator(np.random.PCG64(101))
    nvar = 10
    card = np.array([2, 2, 3, 3, 2, 3, 2, 3, 2, 3], np.int64)
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = card
    nedge = 48
    fmap = np.zeros(nedge, FactorToVar)
    # member ids avoid variable 9 so that "var_samp not in the factor" is exercised too
    fmap["vid"] = rng.integers(0, 9, nedge)
    fmap["dense_equal_to"] = rng.integers(0, 3, nedge)
    funcs = sorted(ref_inf.FACTORS.values()) + [5, 99]        # two undefined ids
    facs = []
    for fn in funcs:
        for arity in (1, 2, 3, 4):
            for rep in range(3):
                # offsets < 8 keep the literal head index (edge index) a valid variable id
                off = int(rng.integers(0, 7)) if rep < 2 else int(rng.integers(8, nedge - 4))
                facs.append((fn, arity, off))
    factor = np.zeros(len(facs), Factor)
    for i, (fn, arity, off) in enumerate(facs):
        factor[i]["factorFunction"] = fn
        factor[i]["arity"] = arity
        factor[i]["ftv_offset"] = off
        factor[i]["featureValue"] = 1.0
    nstate = 6
    states = np.stack([(rng.random((nstate,)) * c).astype(np.int64) for c in card], axis=1)
    rows = []
    for fid in range(len(facs)):
        for s in range(nstate):
            vv = states[s][None, :].copy()
            for var_samp in range(nvar):
                for value in range(int(card[var_samp])):
                    try:
                        r = quiet(ref_inf.eval_factor, fid, var_samp, value, 0, variable, factor,
                                  fmap, vv)
                        status = 0
                    except NotImplementedError:
                        r, status = 0.0, 1
                    except IndexError:
                        r, status = 0.0, 2
                    rows.append((fid, s, var_samp, value, status, float(r)))
    rows = np.array(rows, np.float64)
    np.savez_compressed(os.path.join(OUT, "g1_eval_factor.npz"), variable=variable, factor=factor,
                        fmap=fmap, states=states, cases=rows)
    print("G1: %d factors, %d cases" % (len(facs), len(rows)))


# --------------------------------------------------------------------------------------------
# small graphs shared by G2/G3/G4
# --------------------------------------------------------------------------------------------
def coin_graph_dir():
    """tests/golden/test_coin = the reference's test/ fixture data (graph.meta trimmed to the
    four fields the Meta dtype has; modern numpy rejects the 8-field original)."""
    d = os.path.join(OUT, "test_coin")
    os.makedirs(d, exist_ok=True)
    for name in ("graph.weights", "graph.variables", "graph.factors"):
        shutil.copyfile(os.path.join(REF, "test", name), os.path.join(d, name))
    with open(os.path.join(REF, "test", "graph.meta")) as f:
        fields = f.read().strip().split(",")
    with open(os.path.join(d, "graph.meta.orig"), "w") as f:
        f.write(",".join(fields))
    with open(os.path.join(d, "graph.meta"), "w") as f:
        f.write(",".join(fields[:4]))
    return d


def mixed_categorical_graph(seed=5):
    """12 variables (booleans, dataType-1 categoricals, one dataType-0 cardinality-3), factors of
    most boolean and categorical kinds, duplicates inside a factor, 3 weights."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nvar = 12
    variable = np.zeros(nvar, Variable)
    dt = np.array([0, 0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1], np.int16)
    card = np.array([2, 2, 3, 4, 2, 3, 2, 3, 5, 2, 2, 3], np.int64)
    variable["dataType"] = dt
    variable["cardinality"] = card
    variable["isEvidence"] = np.array([0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0], np.int8)
    variable["initialValue"] = (rng.random(nvar) * card).astype(np.int64)
    spec = [  # (function, member ids)
        (4, [0]), (4, [1]), (3, [0, 1]), (1, [0, 4, 6]), (2, [4, 6]), (0, [1, 4, 9]),
        (7, [0, 6, 9, 10]), (8, [4, 9, 10]), (9, [6, 10, 0]),
        (12, [2, 3]), (14, [2, 5, 0]), (15, [8]), (12, [3, 8, 11]), (14, [11, 1]),
        (14, [5, 5, 2]), (3, [9, 9, 10]), (12, [7, 2]), (14, [7, 3]), (4, [7]),
        (12, [8, 11]), (14, [3, 11, 8, 5]),
    ]
    nfac = len(spec)
    factor = np.zeros(nfac, Factor)
    nedge = sum(len(m) for _, m in spec)
    fmap = np.zeros(nedge, FactorToVar)
    e = 0
    for i, (fn, members) in enumerate(spec):
        factor[i]["factorFunction"] = fn
        factor[i]["weightId"] = i % 3
        factor[i]["featureValue"] = [1.0, 0.5, 2.0][i % 3] if i % 5 == 0 else 1.0
        factor[i]["arity"] = len(members)
        factor[i]["ftv_offset"] = e
        for m in members:
            fmap[e]["vid"] = m
            fmap[e]["dense_equal_to"] = int(rng.integers(0, card[m]))
            e += 1
    weight = np.zeros(3, Weight)
    weight["initialValue"] = [0.7, -0.4, 0.25]
    weight["isFixed"] = [False, False, True]
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


def head_quirk_graph():
    """IMPLY_MLN / IMPLY_NATURAL_CAT / IMPLY_MLN_CAT factors laid out so that the reference's
    literal head lookup var_value[l] (inference.py:243,277,292) stays inside the variable
    array and differs from the intended fmap[l].vid lookup."""
    nvar = 16          # >= number of edges, so every literal head index is a valid variable id
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    variable["initialValue"] = [0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1]
    spec = [(13, [5, 7, 9]), (13, [0, 2]), (16, [1, 3, 8]), (17, [4, 6, 11]), (13, [10, 0, 1])]
    nedge = sum(len(m) for _, m in spec)
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(nedge, FactorToVar)
    e = 0
    for i, (fn, members) in enumerate(spec):
        factor[i]["factorFunction"] = fn
        factor[i]["weightId"] = i % 2
        factor[i]["featureValue"] = 1.0
        factor[i]["arity"] = len(members)
        factor[i]["ftv_offset"] = e
        for m in members:
            fmap[e]["vid"] = m
            fmap[e]["dense_equal_to"] = (m + i) % 2
            e += 1
    weight = np.zeros(2, Weight)
    weight["initialValue"] = [0.8, -0.6]
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


def skip_graph():
    """Graph for ``factors_to_skip``.  The reference sizes factor_index by the NON-skipped edges
    but counts slot lengths over every edge (numbskull.py:217, dataloading.py:34-38), so in
    pure-Python mode only factors whose members fall in the last slot can be skipped without
    an IndexError; f4 and f5 are such factors."""
    variable = np.zeros(4, Variable)
    variable["cardinality"] = 2
    spec = [(3, [0, 1]), (1, [1, 2]), (4, [3]), (2, [2, 3]), (4, [3]), (4, [3])]
    nedge = sum(len(m) for _, m in spec)
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(nedge, FactorToVar)
    e = 0
    for i, (fn, members) in enumerate(spec):
        factor[i]["factorFunction"] = fn
        factor[i]["featureValue"] = 1.0
        factor[i]["arity"] = len(members)
        factor[i]["ftv_offset"] = e
        for m in members:
            fmap[e]["vid"] = m
            e += 1
    weight = np.zeros(1, Weight)
    weight["initialValue"] = 0.3
    return weight, variable, factor, fmap, np.zeros(4, np.bool_), nedge


# --------------------------------------------------------------------------------------------
# G2  index build (pins rows a9 / a11)
# --------------------------------------------------------------------------------------------
def g2_index_build(coin_dir):
    out = {}

    def capture(tag, g, factors_to_skip=None):
        ns, fg = load_ref(g, factors_to_skip)
        out.update(graph_arrays(tag + "_in_", g))
        if factors_to_skip is not None:
            out[tag + "_in_factors_to_skip"] = np.asarray(factors_to_skip, np.int64)
        out[tag + "_out_variable"] = fg.variable.copy()
        out[tag + "_out_vmap"] = fg.vmap.copy()
        out[tag + "_out_factor_index"] = fg.factor_index.copy()
        out[tag + "_out_cstart"] = fg.cstart.copy()

    capture("grid4x5", graphgen.ising_grid(4, 5, weight=0.5))
    capture("mixed", mixed_categorical_graph())
    capture("skiplast", skip_graph(), factors_to_skip=[4, 5])
    capture("pairs", graphgen.ising_pairs(6, seed=3))
    capture("lf", graphgen.lf_graph(0.0, [1.0, 0.5], 4, seed=2))

    # file loaders: the coin fixture, and a categorical graph with explicit graph.domains
    ns = ref.NumbSkull(directory=coin_dir, quiet=True)
    quiet(ns.loadFGFromFile)
    fg = ns.factorGraphs[0]
    for name in ("weight", "variable", "factor", "fmap", "vmap", "factor_index", "cstart"):
        out["coin_out_" + name] = getattr(fg, name).copy()

    dom_dir = os.path.join(OUT, "domains_graph")
    g = mixed_categorical_graph(seed=9)
    domains = {2: [3, 7, 11], 3: [0, 5, 6, 9], 8: [10, 20, 30, 40, 50]}
    graphgen.write_graph(dom_dir, g[0], g[1], g[2], g[3], domains=domains)
    ns = ref.NumbSkull(directory=dom_dir, quiet=True)
    quiet(ns.loadFGFromFile)
    fg = ns.factorGraphs[0]
    for name in ("weight", "variable", "factor", "fmap", "vmap", "factor_index", "cstart"):
        out["domains_out_" + name] = getattr(fg, name).copy()
    np.savez_compressed(os.path.join(OUT, "g2_index_build.npz"), **out)
    print("G2: %d arrays" % len(out))


# --------------------------------------------------------------------------------------------
# G3  seeded inference traces (pins rows a1-a3, a8)
# --------------------------------------------------------------------------------------------
def trace_inference(fg, burnin, epochs, sample_evidence):
    vals, counts = [], []
    if burnin:
        quiet(fg.burnIn, burnin, sample_evidence)
    vals.append(fg.var_value[0].copy())
    for _ in range(epochs):
        quiet(fg.inference, 0, 1, sample_evidence)
        vals.append(fg.var_value[0].copy())
        counts.append(fg.count.copy())
    return np.array(vals), np.array(counts)


def g3_inference(coin_dir):
    out = {}
    cases = [
        ("grid4x5_w05", graphgen.ising_grid(4, 5, weight=0.5), 42, 10, 40, True),
        ("grid32_w01", graphgen.ising_grid(32, 32, weight=0.1), 7, 2, 6, True),
        ("grid32_w05", graphgen.ising_grid(32, 32, weight=0.5), 8, 2, 6, True),
        ("mixed", mixed_categorical_graph(), 11, 3, 30, True),
        ("mixed_noev", mixed_categorical_graph(), 12, 3, 30, False),
        ("lf", graphgen.lf_graph(0.3, [1.0, 0.5], 5, seed=4), 13, 2, 20, True),
        ("headquirk", head_quirk_graph(), 14, 2, 30, True),
    ]
    for tag, g, seed, burn, epochs, se in cases:
        ns, fg = load_ref(g)
        seed_all(seed)
        vals, counts = trace_inference(fg, burn, epochs, se)
        out.update(graph_arrays(tag + "_in_", g))
        out[tag + "_seed"] = np.int64(seed)
        out[tag + "_burn"] = np.int64(burn)
        out[tag + "_sample_evidence"] = np.int64(se)
        out[tag + "_var_value"] = vals
        out[tag + "_count"] = counts
    # 4x5 grid, 1000 epochs: the BASELINE.md section 2 run
    ns, fg = load_ref(graphgen.ising_grid(4, 5, weight=0.5))
    seed_all(42)
    quiet(fg.inference, 10, 1000, True)
    out["grid4x5_long_count"] = fg.count.copy()
    out["grid4x5_long_marginals"] = np.asarray(fg.marginals).copy()
    np.savez_compressed(os.path.join(OUT, "g3_inference.npz"), **out)
    print("G3: %d arrays" % len(out))


# --------------------------------------------------------------------------------------------
# G4  seeded learning traces (pins rows a5-a7)
# --------------------------------------------------------------------------------------------
def trace_learning(fg, epochs, stepsize, decay, reg, reg_param, trunc, lne):
    ws, vv, ve = [fg.weight_value[0].copy()], [fg.var_value[0].copy()], [fg.var_value_evid[0].copy()]
    for _ in range(epochs):
        quiet(fg.learn, 0, 1, stepsize, decay, reg, reg_param, trunc, learn_non_evidence=lne)
        stepsize *= decay
        ws.append(fg.weight_value[0].copy())
        vv.append(fg.var_value[0].copy())
        ve.append(fg.var_value_evid[0].copy())
    return np.array(ws), np.array(vv), np.array(ve)


def g4_learning(coin_dir):
    out = {}
    graphs = {
        "pairs": graphgen.ising_pairs(40, seed=3),
        "mixed": mixed_categorical_graph(),
        "lf": graphgen.lf_graph(0.0, [1.0, 0.5], 10, seed=2),
    }
    for tag, g in graphs.items():
        out.update(graph_arrays(tag + "_in_", g))
    k = 0
    for tag, g in graphs.items():
        for reg in (0, 1, 2):
            for lne in (False, True):
                for trunc in ((1, 3) if reg == 1 else (1,)):
                    name = "%s_r%d_l%d_k%d" % (tag, reg, int(lne), trunc)
                    ns, fg = load_ref(g)
                    seed = 100 + k
                    k += 1
                    seed_all(seed)
                    ws, vv, ve = trace_learning(fg, 8, 0.05, 0.9, reg, 0.02, trunc, lne)
                    out[name + "_seed"] = np.int64(seed)
                    out[name + "_weights"] = ws
                    out[name + "_var_value"] = vv
                    out[name + "_var_value_evid"] = ve
    # config #1: the reference CLI run  `numbskull test -l 10 -i 10`  (README.md:27)
    d = tempfile.mkdtemp()
    ns = quiet(ref.numbskull.load, [coin_dir, "-l", "10", "-i", "10", "-o", d, "--quiet"])
    seed_all(1234)
    quiet(ns.learning)
    quiet(ns.inference)
    fg = ns.factorGraphs[0]
    out["coin_cli_weights"] = fg.weight_value[0].copy()
    out["coin_cli_count"] = fg.count.copy()
    out["coin_cli_var_value"] = fg.var_value[0].copy()
    out["coin_cli_var_value_evid"] = fg.var_value_evid[0].copy()
    with open(os.path.join(d, "inference_result.out.text")) as f:
        out["coin_cli_probs_text"] = np.array(f.read())
    with open(os.path.join(d, "inference_result.out.weights.text")) as f:
        out["coin_cli_weights_text"] = np.array(f.read())
    shutil.rmtree(d)
    # the reference's own test.py parameters, single-threaded so that it is reproducible
    ns = quiet(ref.numbskull.load, [coin_dir, "-l", "100", "-i", "100", "-s", "0.01",
                                    "--regularization", "2", "-r", "0.1", "--quiet"])
    seed_all(99)
    quiet(ns.learning, 0, False)
    quiet(ns.inference, 0, False)
    out["coin_testpy_weights"] = ns.factorGraphs[0].weight_value[0].copy()
    out["coin_testpy_count"] = ns.factorGraphs[0].count.copy()
    np.savez_compressed(os.path.join(OUT, "g4_learning.npz"), **out)
    print("G4: %d arrays" % len(out))


# --------------------------------------------------------------------------------------------
# G7  MT19937 streams as the reference consumes them (np.random.rand / random.random)
# --------------------------------------------------------------------------------------------
def g7_rng():
    out = {}
    for s in (0, 1, 42, 1234, 20240601, 2 ** 32 - 1):
        np.random.seed(s)
        out["np_%d" % s] = np.array([np.random.rand() for _ in range(700)])
        random.seed(s)
        out["py_%d" % s] = np.array([random.random() for _ in range(700)])
    np.savez_compressed(os.path.join(OUT, "g7_rng.npz"), **out)
    print("G7: %d arrays" % len(out))


def main():
    os.makedirs(OUT, exist_ok=True)
    coin_dir = coin_graph_dir()
    g1_eval_factor()
    g2_index_build(coin_dir)
    g3_inference(coin_dir)
    g4_learning(coin_dir)
    g7_rng()


if __name__ == "__main__":
    main()
