"""GPU parity tests: the HIP path, called through the C-ABI, against

  (1) the REFERENCE's own traces (tests/golden, captured from HazyResearch/numbskull) with the
      sequential scan -- same trajectory, bit-exact on values / tallies / weights;
  (2) the CPU oracle's device mode with the chromatic scan -- bit-exact on values, tallies and
      weights for every factor function, regulariser and data type;
  (3) exact marginals by enumeration -- |delta marginal| within the sampling tolerance stated
      in each test.
"""

import os

import numpy as np
import pytest

from conftest import graph_from, GOLDEN
from numbskull_amd import _lib, graphgen
from util import (orc, quiet, session, oracle_of, phases_from_colors, check_coloring,
                  exact_marginals)
import numbskull_amd

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------
# device primitives
# ------------------------------------------------------------------------------------------
def test_device_exp_equals_oracle_exp_bitwise():
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-50, 50, 400000), rng.uniform(-745.2, 709.8, 100000),
                        rng.normal(0, 2, 400000),
                        [0.0, -0.0, 1.0, -1.0, 709.782712893384, 709.79, -745.13, -745.14, -708.4,
                         -720.0, np.inf, -np.inf, 1e-310, 5e-324]])
    y = np.empty_like(x)
    _lib.check(_lib.lib().nsk_selftest_exp(0, _lib.ptr(x), _lib.ptr(y), len(x)))
    want = orc.exp_det(x[:20000])
    assert np.array_equal(y[:20000].view(np.uint64), want.view(np.uint64))
    tail = orc.exp_det(x[-5000:])
    assert np.array_equal(y[-5000:].view(np.uint64), tail.view(np.uint64))
    # and within 1 ulp of libm everywhere
    ref = np.exp(x)
    ok = np.abs(y - ref) <= np.spacing(ref)
    assert np.all(ok | (np.isinf(ref) & np.isinf(y)))


def test_device_philox_equals_oracle():
    n = 4096
    out = np.zeros(4 * n, np.uint32)
    seed, sweep = 0x0123456789abcdef, 0x00000007deadbeef
    _lib.check(_lib.lib().nsk_selftest_philox(0, seed, sweep, 1, n, _lib.ptr(out)))
    for i in (0, 1, 2, 63, 64, 1000, n - 1):
        want = orc.philox(seed & 0xffffffff, seed >> 32, i, 1, sweep & 0xffffffff, sweep >> 32)
        assert out[4 * i:4 * i + 4].tolist() == want


# ------------------------------------------------------------------------------------------
# (1) sequential scan == the reference's own trajectory
# ------------------------------------------------------------------------------------------
G3_TAGS = ["grid4x5_w05", "grid32_w01", "mixed", "mixed_noev", "lf", "headquirk"]


@pytest.mark.parametrize("tag", G3_TAGS)
def test_sequential_inference_matches_reference_trace(golden, tag):
    z = golden("g3_inference.npz")
    seed, burn = int(z[tag + "_seed"]), int(z[tag + "_burn"])
    se = bool(z[tag + "_sample_evidence"])
    ns, fg = session(graph_from(z, tag), scan="sequential", seed=seed)
    vals, counts = z[tag + "_var_value"], z[tag + "_count"]
    fg.burnIn(burn, se)
    assert np.array_equal(fg.var_value[0], vals[0])
    for e in range(len(counts)):
        fg.inference(0, 1, se)
        assert np.array_equal(fg.var_value[0], vals[e + 1]), (tag, e)
        assert np.array_equal(fg.count, counts[e]), (tag, e)
    assert np.array_equal(fg.marginals, counts[-1] / 1.0)


def _g4_cases():
    for tag in ("pairs", "mixed", "lf"):
        for reg in (0, 1, 2):
            for lne in (0, 1):
                for k in ((1, 3) if reg == 1 else (1,)):
                    yield tag, reg, lne, k


@pytest.mark.parametrize("tag,reg,lne,trunc", list(_g4_cases()))
def test_sequential_learning_matches_reference_trace(golden, tag, reg, lne, trunc):
    z = golden("g4_learning.npz")
    name = "%s_r%d_l%d_k%d" % (tag, reg, lne, trunc)
    ns, fg = session(graph_from(z, tag), scan="sequential", seed=int(z[name + "_seed"]))
    ws, vvs, ves = z[name + "_weights"], z[name + "_var_value"], z[name + "_var_value_evid"]
    # all 8 epochs in ONE call: the decay loop runs inside the library (factorgraph.py:206)
    fg.learn(0, len(ws) - 1, 0.05, 0.9, reg, 0.02, trunc, learn_non_evidence=bool(lne))
    assert np.array_equal(fg.var_value[0], vvs[-1])
    assert np.array_equal(fg.var_value_evid[0], ves[-1])
    assert np.array_equal(fg.weight_value[0], ws[-1]), (fg.weight_value[0], ws[-1])


def test_config1_cli_run_matches_reference(golden, tmp_path):
    """BASELINE config #1: `numbskull test -l 10 -i 10` (README.md:27), seed 1234."""
    z = golden("g4_learning.npz")
    coin = os.path.join(GOLDEN, "test_coin")
    ns = quiet(numbskull_amd.numbskull.load,
               [coin, "-l", "10", "-i", "10", "-o", str(tmp_path), "--quiet",
                "--scan", "sequential", "--seed", "1234"])
    quiet(ns.learning)
    quiet(ns.inference)
    fg = ns.factorGraphs[0]
    assert np.array_equal(fg.weight_value[0], z["coin_cli_weights"])
    assert abs(fg.weight_value[0][0] - 0.330407) < 1e-6
    assert np.array_equal(fg.count, z["coin_cli_count"])
    assert fg.count.tolist() == [7, 6, 8, 4, 7, 7, 4, 6, 7, 8, 6, 5, 6, 7, 8, 8, 10, 7]
    assert np.array_equal(fg.var_value[0], z["coin_cli_var_value"])
    assert np.array_equal(fg.var_value_evid[0], z["coin_cli_var_value_evid"])
    assert (tmp_path / "inference_result.out.text").read_text() == str(z["coin_cli_probs_text"])
    assert (tmp_path / "inference_result.out.weights.text").read_text() == \
        str(z["coin_cli_weights_text"])


def test_reference_test_py_parameters(golden):
    """The reference's test.py run (-l 100 -i 100 -s 0.01 --regularization 2 -r 0.1), one thread."""
    z = golden("g4_learning.npz")
    coin = os.path.join(GOLDEN, "test_coin")
    ns = quiet(numbskull_amd.numbskull.load,
               [coin, "-l", "100", "-i", "100", "-s", "0.01", "--regularization", "2", "-r", "0.1",
                "--quiet", "--scan", "sequential", "--seed", "99"])
    quiet(ns.learning, 0, False)
    quiet(ns.inference, 0, False)
    assert np.array_equal(ns.factorGraphs[0].weight_value[0], z["coin_testpy_weights"])
    assert np.array_equal(ns.factorGraphs[0].count, z["coin_testpy_count"])


# ------------------------------------------------------------------------------------------
# (2) chromatic scan == oracle device mode, bit for bit
# ------------------------------------------------------------------------------------------
def _small_graphs(golden):
    z3, z4 = golden("g3_inference.npz"), golden("g4_learning.npz")
    return {
        "grid4x5": (graph_from(z3, "grid4x5_w05"), False),
        "grid32": (graph_from(z3, "grid32_w05"), False),
        "mixed": (graph_from(z3, "mixed"), False),
        "lf": (graph_from(z3, "lf"), False),
        "headquirk": (graph_from(z3, "headquirk"), False),
        "headquirk_vid": (graph_from(z3, "headquirk"), True),
        "pairs": (graph_from(z4, "pairs"), False),
        "grid57x33": (graphgen.ising_grid(57, 33, weight=0.3), False),
        "lr3000": (graphgen.mixed_lr_graph(3000, seed=5, nweights=40), True),
        # > 256 weights: gradients go through global atomics instead of per-block LDS tables
        "lr_manyw": (graphgen.mixed_lr_graph(3000, seed=6, nweights=1500), True),
        # other members drawn from [v - 2, v + 2]: a fifth of them is the variable itself -- body
        # member AND head of IMPLY_MLN / IMPLY_MLN_CAT, own edges with different dense_equal_to --
        # the two-role entries of the general tiles (nsk_compile.cpp general_words)
        "lr_selfdup": (graphgen.mixed_lr_graph(3000, seed=8, nweights=40, window=2), True),
        "pairs_manyw": (_pairs_many_weights(), False),
        # one weight per factor: tiles share a word layout but not weights (shape tiles)
        "boolw": (_boolw(), False),
        # hub variables (factor lists >= 128 entries): one wave per variable
        "hubs": (_hub_graph(), False),
        "lr_bigcard": (_big_cardinality_graph(), False),
        # every function of the general tiles (kind 6) on dataType-0 and -1 variables of
        # cardinality 2..8, with the literal and the by-vid head lookup
        "gencat": (_general_tile_graph(), False),
        "gencat_vid": (_general_tile_graph(), True),
        # weight table beyond 4 MB: general tiles read materialised weight rows in inference
        "gencat_bigw": (_general_tile_graph(nweight=600000), True),
        "gencat_i32": (_general_tile_graph(wide_values=True), True),
        # arity <= 4: every entry has at most 3 other members, so these colours are laid out as
        # entry-parallel groups (k_gibbs_ep / k_learn_ep) -- all twelve general-tile functions, repeated
        # members, both head lookups, int8 and int32 values
        "gencat4": (_general_tile_graph(maxarity=4), False),
        "gencat4_vid": (_general_tile_graph(maxarity=4), True),
        "gencat4_i32": (_general_tile_graph(wide_values=True, maxarity=4), True),
    }


def _general_tile_graph(nweight=30, wide_values=False, maxarity=5):
    from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar
    rng = np.random.default_rng(23)
    nvar, nfactor = 6000, 1700
    variable = np.zeros(nvar, Variable)
    card = rng.integers(2, 9, nvar)
    card[rng.random(nvar) < 0.03] = 30                 # too large for a general tile: generic path
    if wide_values:
        card[::997] = 200                              # values no longer fit int8: the int32 kernels
    variable["cardinality"] = card
    variable["dataType"] = rng.random(nvar) < 0.5
    variable["isEvidence"] = rng.random(nvar) < 0.5
    variable["initialValue"] = (rng.random(nvar) * card).astype(np.int64)
    funcs = np.array([-1, 0, 1, 2, 3, 4, 12, 13, 14, 15, 16, 17])
    arity = rng.integers(1, maxarity + 1, nfactor)
    off = np.cumsum(arity) - arity
    nedge = int(arity.sum())
    assert nedge < nvar                                # literal head lookup reads var_value[edge index]
    fmap = np.zeros(nedge, FactorToVar)
    for f in range(nfactor):
        base = int(rng.integers(0, nvar // 3))            # dense third, the rest without factors
        members = base + rng.choice(64 if f % 8 else 4, size=arity[f], replace=bool(f % 8 == 0))   # some repeated members
        fmap["vid"][off[f]:off[f] + arity[f]] = members
    fmap["dense_equal_to"] = (rng.random(nedge) * card[fmap["vid"]]).astype(np.int64)
    factor = np.zeros(nfactor, Factor)
    factor["factorFunction"] = funcs[rng.integers(0, len(funcs), nfactor)]
    factor["weightId"] = rng.integers(0, nweight, nfactor)
    factor["featureValue"] = 1.0
    factor["arity"] = arity
    factor["ftv_offset"] = off
    weight = np.zeros(nweight, Weight)
    weight["initialValue"] = rng.normal(0, 0.3, nweight)
    weight["isFixed"][::5] = True
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


def _hub_graph():
    """three hubs: a boolean one under 300 OR / EQUAL / IMPLY_MLN factors, a dataType-1 categorical
    one (cardinality 5) under 400 AND_CAT / OR_CAT factors, and a data-programming label under 200
    labelling-function accuracy factors; leaves carry ISTRUE priors; 7 free weights"""
    from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar
    rng = np.random.default_rng(17)
    nleaf_b, nleaf_c, nlf = 300, 400, 200
    nvar = 3 + nleaf_b + nleaf_c + nlf
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    variable[1]["cardinality"] = 5
    variable[1]["dataType"] = 1
    lf0 = 3 + nleaf_b + nleaf_c
    variable["cardinality"][lf0:] = 3
    variable["isEvidence"] = rng.random(nvar) < 0.6
    variable["isEvidence"][:3] = [0, 1, 0]
    variable["initialValue"] = (rng.random(nvar) * variable["cardinality"]).astype(np.int64)
    spec = []
    for i in range(nleaf_b):
        leaf = 3 + i
        fn = (1, 3, 13)[i % 3]
        spec.append((fn, [leaf, 0] if i % 2 else [0, leaf], [0, 0]))
        spec.append((4, [leaf], [0]))
    for i in range(nleaf_c):
        leaf = 3 + nleaf_b + i
        fn = (12, 14)[i % 2]
        spec.append((fn, [1, leaf], [int(rng.integers(0, 5)), int(rng.integers(0, 2))]))
    for i in range(nlf):
        spec.append((21, [2, lf0 + i], [0, 0]))
    spec.append((18, [2], [0]))
    nedge = sum(len(m) for _, m, _ in spec)
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(nedge, FactorToVar)
    e = 0
    for i, (fn, members, deos) in enumerate(spec):
        factor[i] = (fn, i % 7, 1.0, len(members), e)
        for m, dq in zip(members, deos):
            fmap[e] = (m, dq)
            e += 1
    weight = np.zeros(7, Weight)
    weight["initialValue"] = rng.normal(0, 0.05, 7)
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), nedge


def _boolw():
    g = list(graphgen.boolean_weighted_graph(6000, seed=4, max_arity=5, factors_per_var=1.0))
    w = g[0].copy()
    w["isFixed"] = False
    var = g[1].copy()
    rng = np.random.default_rng(8)
    var["isEvidence"] = rng.random(len(var)) < 0.5
    var["initialValue"] = rng.integers(0, 2, len(var))
    g[0], g[1] = w, var
    return tuple(g)


def _pairs_many_weights():
    """pair model with one weight per pair for the EQUAL factors (600 weights): uniform tiles are
    impossible (every lane has its own weight id) -> per-lane-header tiles in learning"""
    g = list(graphgen.ising_pairs(300, seed=2))
    fac = g[2].copy()
    eq = np.nonzero(fac["factorFunction"] == 3)[0]
    fac["weightId"][eq] = 3 + np.arange(len(eq))
    from numbskull_amd.numbskulltypes import Weight
    g[0] = np.zeros(3 + len(eq), Weight)
    g[2] = fac
    return tuple(g)


def _big_cardinality_graph():
    """cardinalities 20 and 200: exercises the two-pass draw and the int32 value type"""
    g = list(graphgen.mixed_lr_graph(400, seed=3, nweights=9))
    var = g[1].copy()
    cat = np.nonzero(var["dataType"] == 1)[0]
    var["cardinality"][cat[::2]] = 20
    var["cardinality"][cat[1::2]] = 200
    var["initialValue"] = np.minimum(var["initialValue"], var["cardinality"] - 1)
    fm = g[3].copy()
    fm["dense_equal_to"] = fm["dense_equal_to"] % var["cardinality"][fm["vid"]]
    g[1], g[3] = var, fm
    # the head lookup quirk needs edge index < nvar: keep only non-IMPLY functions
    fac = g[2].copy()
    fac["factorFunction"][fac["factorFunction"] == 13] = 1
    fac["factorFunction"][fac["factorFunction"] == 17] = 14
    g[2] = fac
    return tuple(g)


GRAPHS = ["grid4x5", "grid32", "mixed", "lf", "headquirk", "headquirk_vid", "pairs", "grid57x33",
          "lr3000", "lr_bigcard", "lr_manyw", "pairs_manyw", "boolw", "hubs", "gencat", "gencat_vid",
          "gencat_bigw", "gencat_i32", "lr_selfdup", "gencat4", "gencat4_vid", "gencat4_i32"]


@pytest.mark.parametrize("name", GRAPHS)
@pytest.mark.parametrize("sample_evidence", [True, False])
def test_chromatic_inference_equals_oracle(golden, name, sample_evidence):
    g, hbv = _small_graphs(golden)[name]
    ns, fg = session(g, seed=77, head_by_vid=hbv)
    og = oracle_of(fg, hbv)
    color = fg.colors()
    check_coloring(fg, color, hbv)
    order, ps = phases_from_colors(color)
    vv, _, wv, cnt = og.initial_state()
    sweep = 0
    fg.burnIn(3, sample_evidence)
    for _ in range(3):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 77, sweep, sample_evidence, burnin=True) == 0
        sweep += 1
    assert np.array_equal(fg.var_value[0], vv)
    for rounds in (1, 4):
        fg.inference(0, rounds, sample_evidence)
        for _ in range(rounds):
            assert og.gibbs_dev(order, ps, vv, wv, cnt, 77, sweep, sample_evidence) == 0
            sweep += 1
        assert np.array_equal(fg.var_value[0], vv), name
        assert np.array_equal(fg.count, cnt), name


@pytest.mark.parametrize("name", ["mixed", "lf", "pairs", "grid32", "lr3000", "lr_bigcard",
                                  "headquirk", "lr_manyw", "pairs_manyw", "boolw", "hubs", "gencat",
                                  "gencat_vid", "gencat_i32", "lr_selfdup", "gencat4", "gencat4_vid",
                                  "gencat4_i32"])
@pytest.mark.parametrize("reg,trunc", [(0, 1), (1, 1), (1, 3), (2, 1)])
@pytest.mark.parametrize("lne", [False, True])
def test_chromatic_learning_equals_oracle(golden, name, reg, trunc, lne):
    g, hbv = _small_graphs(golden)[name]
    if name == "grid32":                      # make the grid learnable: free weight, evidence
        w = g[0].copy()
        w["isFixed"] = False
        rng = np.random.default_rng(1)
        g = graphgen.ising_grid(32, 32, weight=0.2, fixed=False, two_weights=True,
                                evidence=rng.integers(0, 2, 32 * 32))
    ns, fg = session(g, seed=5, head_by_vid=hbv)
    og = oracle_of(fg, hbv)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    step, decay, sweep = 0.01, 0.9, 0
    for chunk in (1, 3):
        fg.learn(0, chunk, step, decay, reg, 0.05, trunc, learn_non_evidence=lne)
        assert og.learn_call(order, ps, vv, ve, wv, chunk, step, decay, reg, 0.05, trunc, lne, 5, sweep) == 0
        for _ in range(chunk):
            step *= decay
            sweep += 1
        assert np.array_equal(fg.var_value[0], vv), name
        assert np.array_equal(fg.var_value_evid[0], ve), name
        assert np.array_equal(fg.weight_value[0], wv), (name, fg.weight_value[0], wv)
    learns = lne or np.any(og.variable["isEvidence"] == 1)
    assert np.any(wv != og.weight["initialValue"]) or og.weight["isFixed"].all() or not learns


@pytest.mark.parametrize("name,no_general,no_heavy", [
    ("lr3000", 1, 0), ("gencat", 1, 0), ("lr3000", 1, 1), ("gencat", 1, 1), ("gencat_vid", 1, 1),
    ("gencat_bigw", 0, 0), ("mixed", 0, 1), ("lf", 0, 1), ("lr_bigcard", 0, 1), ("headquirk", 1, 1), ("hubs", 0, 1)])
def test_generic_kernels_alone(golden, name, no_general, no_heavy, monkeypatch):
    """The layout heuristics send small test graphs to the tile and the wave-per-variable kernels;
    NSK_NO_GENERAL / NSK_NO_HEAVY (diagnostic switches read at graph creation) keep the variables
    on the wave-per-variable / the one-lane generic kernels instead: same results."""
    if no_general:
        monkeypatch.setenv("NSK_DIAG", "1")
        monkeypatch.setenv("NSK_NO_GENERAL", "1")
    if no_heavy:
        monkeypatch.setenv("NSK_DIAG", "1")
        monkeypatch.setenv("NSK_NO_HEAVY", "1")
    g, hbv = _small_graphs(golden)[name]
    ns, fg = session(g, seed=5, head_by_vid=hbv)
    assert fg.info()["ngeneric"] > (1000 if no_general and name != "headquirk" else 0)
    og = oracle_of(fg, hbv)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 2, 0.01, 0.9, 2, 0.05, 1, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 2, 0.01, 0.9, 2, 0.05, 1, True, 5, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv)
    fg.inference(0, 3, True)
    for sweep in range(2, 5):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 5, sweep, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


@pytest.mark.parametrize("switch", ["NSK_ONE_ACC", "NSK_NO_KSTAT"])
@pytest.mark.parametrize("name", ["lr_manyw", "gencat_bigw"])
def test_learning_accumulator_fallbacks(golden, name, switch, monkeypatch):
    """NSK_ONE_ACC: one copy of the global gradient accumulators updated with agent-scope atomics -- what
    an architecture other than gfx942 / gfx950 gets instead of the XCD-private copies; NSK_NO_KSTAT:
    every visit goes through the accumulators instead of the structural visit counts.  Same weights."""
    g, hbv = _small_graphs(golden)[name]
    if switch == "NSK_ONE_ACC":
        # the product configuration on an MI355X: the device passed nsk_graph_create's self-test of the
        # XCD-private accumulators (k_xcd_selftest), so a many-weight graph keeps 8 copies -- while a copy
        # fits its XCD's L2 (up to 2^18 weights); a larger table has one copy in the product as well
        ns0, fg0 = session(g, seed=5, head_by_vid=hbv)
        assert fg0.info()["acc_copies"] & 15 == (8 if len(g[0]) <= 1 << 18 else 1)
        fg0.close()
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv(switch, "1")
    ns, fg = session(g, seed=5, head_by_vid=hbv)
    if switch == "NSK_ONE_ACC":
        assert fg.info()["acc_copies"] == 1
    og = oracle_of(fg, hbv)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    fg.learn(0, 3, 0.01, 0.9, 2, 0.05, 1, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 3, 0.01, 0.9, 2, 0.05, 1, True, 5, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv)


@pytest.mark.parametrize("name", ["lr_manyw", "pairs_manyw", "boolw", "gencat_bigw"])
@pytest.mark.parametrize("reg", [1, 2])
def test_learning_with_unpacked_accumulators(golden, name, reg, monkeypatch):
    """Graphs with integer gradients carry the visit count in the low half of the 64-bit gradient
    accumulator (one atomic per visit); NSK_NO_PACKED keeps the separate counters that graphs with
    fractional gradients use: same weights."""
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_NO_PACKED", "1")
    g, hbv = _small_graphs(golden)[name]
    ns, fg = session(g, seed=5, head_by_vid=hbv)
    og = oracle_of(fg, hbv)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    fg.learn(0, 3, 0.01, 0.9, reg, 0.05, 2, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 3, 0.01, 0.9, reg, 0.05, 2, True, 5, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv)


@pytest.mark.parametrize("bins", ["xcd", "agent"])
@pytest.mark.parametrize("evidence", ["all", "half"])
def test_learning_table_kernel_with_a_tiny_resident_grid(monkeypatch, evidence, bins):
    """k_learn_seg_tab's waves walk several trips each -- across segment boundaries, with the
    gradient counts carried in scalar registers until the slot program changes -- when the grid
    is smaller than the work: NSK_LEARN_GRID_CAP=8 forces that on a 128x128 grid (interior and
    border segments, with and without the evidence chain's own draw): same samples and weights as
    the oracle.  bins: the blocks' partial sums go to bins of their own XCD (workgroup-scope adds in
    that L2, the gfx950 path) or, with NSK_ONE_ACC, to bins shared across XCDs (agent-scope adds)."""
    from numbskull_amd import graphgen
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_LEARN_GRID_CAP", "8" if bins == "xcd" else "64")
    if bins == "agent":
        monkeypatch.setenv("NSK_ONE_ACC", "1")
    rng = np.random.Generator(np.random.PCG64(11))
    ev = rng.integers(0, 2, 128 * 128)
    g = graphgen.ising_grid(128, 128, weight=0.0, fixed=False, two_weights=True, evidence=ev)
    if evidence == "half":
        g[1]["isEvidence"] = rng.random(128 * 128) < 0.5
    ns, fg = session(g, seed=9)
    assert fg.info()["ztab_entries"] > 0
    assert fg.info()["acc_copies"] == (17 if bins == "xcd" else 1)     # LDS sums; bins private to XCDs or not
    og = oracle_of(fg, False)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 3, 0.001, 0.9, 2, 0.01, 1, learn_non_evidence=(evidence == "half"))
    assert og.learn_call(order, ps, vv, ve, wv, 3, 0.001, 0.9, 2, 0.01, 1, evidence == "half", 9, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv)


@pytest.mark.parametrize("cap", ["8", "24"])
def test_inference_table_kernel_with_a_tiny_resident_grid(monkeypatch, cap):
    """k_gibbs_seg_tab's waves walk several tile pairs each, across segment boundaries, when the
    grid is smaller than the work (what the 7-blocks-per-CU grid does on the 10M / 40M grids):
    NSK_TAB_GRID_CAP forces that on a 128x128 grid with an evidence border -- values and tallies
    as the oracle's, eager launches and the captured sweep sequence alike (17 + 3 sweeps)."""
    from numbskull_amd import graphgen
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_TAB_GRID_CAP", cap)
    g = graphgen.ising_grid(128, 128, weight=0.2)
    rng = np.random.Generator(np.random.PCG64(5))
    g[1]["isEvidence"] = rng.random(128 * 128) < 0.1
    ns, fg = session(g, seed=21)
    assert fg.info()["ztab_entries"] > 0
    og = oracle_of(fg, False)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    fg.inference(3, 17, False)
    for s in range(20):
        og.gibbs_dev(order, ps, vv, wv, cnt, 21, s, False, burnin=s < 3)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_general_tiles_at_scale():
    """200 000-variable mixed LR graph: full tiles of every layout (up to 12 entries x 5 words),
    both launches (binary / categorical), thousands of wave-per-variable leftovers."""
    g = graphgen.mixed_lr_graph(200000, seed=11)
    ns, fg = session(g, seed=9, head_by_vid=True)
    info = fg.info()
    assert info["nfast"] > 190000 and 0 < info["ngeneric"] < 10000
    og = oracle_of(fg, True)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 2, 0.001, 0.9, 2, 0.01, 1, learn_non_evidence=False)
    assert og.learn_call(order, ps, vv, ve, wv, 2, 0.001, 0.9, 2, 0.01, 1, False, 9, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv)
    fg.inference(0, 3, True)
    for sweep in range(2, 5):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 9, sweep, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_self_duplicate_members_stay_on_the_tiles(golden):
    """A variable that is body member and head of one of its own factors, or whose own edges carry
    different dense_equal_to values, is a general-tile lane (two-role entries), not a wave-per-variable
    straggler: on the window-2 LR graph a third of the variables are of that kind."""
    g, hbv = _small_graphs(golden)["lr_selfdup"]
    w, v, f, fm = g[0], g[1], g[2], g[3]
    fac_of_edge = np.repeat(np.arange(len(f)), f["arity"])
    head = fm["vid"][f["ftv_offset"] + f["arity"] - 1]
    is_head_edge = np.arange(len(fm)) == (f["ftv_offset"] + f["arity"] - 1)[fac_of_edge]
    selfdup = np.unique(fm["vid"][(fm["vid"] == head[fac_of_edge]) & ~is_head_edge])
    assert len(selfdup) > 500
    ns, fg = session(g, seed=1, head_by_vid=hbv)
    info = fg.info()
    assert info["ngeneric"] < 200, info            # (OR_CAT naming two of > 2 values stays generic)


def test_fast_and_generic_paths_are_both_exercised(golden):
    graphs = _small_graphs(golden)
    ns, fg = session(graphs["grid57x33"][0])
    assert fg.info()["nfast"] == 57 * 33 and fg.info()["ngeneric"] == 0
    ns, fg = session(graphs["lr3000"][0], head_by_vid=True)
    info = fg.info()
    assert info["nfast"] > 2000 and info["ngeneric"] > 0     # general tiles take most of it
    ns, fg = session(graphs["mixed"][0])
    assert fg.info()["ngeneric"] > 0
    ns, fg = session(graphs["gencat"][0])
    assert fg.info()["nfast"] > 1500 and fg.info()["ngeneric"] > 50


def test_learning_then_inference_continue_from_state(golden):
    """learning leaves var_value where inference picks up and host-side edits
    between calls are honoured (the distributed reference patches arrays in place)."""
    g, _ = _small_graphs(golden)["mixed"]
    ns, fg = session(g, seed=9)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 2, 0.02, 0.95, 2, 0.01, 1)
    assert og.learn_call(order, ps, vv, ve, wv, 2, 0.02, 0.95, 2, 0.01, 1, False, 9, 0) == 0
    # patch state on the host, as numbskull_master.py:213-224 does between epochs
    fg.var_value[0][::2] = 0
    fg.weight_value[0][0] += 0.5
    vv[::2] = 0
    wv[0] += 0.5
    fg.inference(1, 5, True)
    og.gibbs_dev(order, ps, vv, wv, cnt, 9, 2, True, burnin=True)
    for s in range(5):
        og.gibbs_dev(order, ps, vv, wv, cnt, 9, 3 + s, True)
    assert np.array_equal(fg.var_value[0], vv)
    assert np.array_equal(fg.count, cnt)
    fg.clear()
    assert not fg.count.any()


def test_ghost_variables_are_skipped(golden):
    """isEvidence == 4 ("not owned", inference.py:21-23) and own_range both exclude variables."""
    g = list(_small_graphs(golden)["grid32"][0])
    var = g[1].copy()
    var["isEvidence"][100:300] = 4
    g[1] = var
    ns, fg = session(tuple(g), seed=3)
    og = oracle_of(fg)
    color = fg.colors()
    assert np.all(color[100:300] == -1) and np.all(color[:100] >= 0)
    order, ps = phases_from_colors(color)
    vv, _, wv, cnt = og.initial_state()
    fg.inference(0, 6, True)
    for s in range(6):
        og.gibbs_dev(order, ps, vv, wv, cnt, 3, s, True)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    assert not fg.count[100:300].any()


# ------------------------------------------------------------------------------------------
# (3) statistics: sampled marginals vs exact enumeration (RNG-independent)
# ------------------------------------------------------------------------------------------
def _boolean_zoo():
    """9 boolean variables under every boolean factor function (dataType 0: the Gibbs conditional
    is the exact conditional of the joint, so enumeration gives the stationary marginals)."""
    from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar
    spec = [(4, [0]), (3, [0, 1]), (1, [1, 2, 3]), (2, [2, 4]), (0, [3, 5, 6]), (7, [4, 5, 7]),
            (8, [6, 7, 8]), (9, [8, 0, 1]), (13, [2, 6, 8]), (3, [5, 7]), (4, [8])]
    nvar = 9
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(sum(len(m) for _, m in spec), FactorToVar)
    e = 0
    for i, (fn, members) in enumerate(spec):
        factor[i] = (fn, i % 4, 1.0, len(members), e)
        for m in members:
            fmap[e]["vid"] = m
            e += 1
    weight = np.zeros(4, Weight)
    weight["initialValue"] = [0.6, -0.5, 0.9, 0.3]
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), e


def test_marginals_match_exact_enumeration(golden):
    """Sampled vs enumerated marginals (RNG-free ground truth) on a 4x3 Ising grid (w=0.5), a
    graph with every boolean factor function, and the data-programming graph: 200k chromatic
    sweeps; a marginal's standard error is <= 0.5/sqrt(N_eff) ~ 2.5e-3 at N_eff ~ N/5, so the
    tolerance is 0.01."""
    lf = graphgen.lf_graph(0.3, [1.0, 0.5], 3, seed=4)
    for g, hbv in ((graphgen.ising_grid(4, 3, weight=0.5), False), (_boolean_zoo(), True),
                   (lf, False)):
        ns, fg = session(g, seed=123, head_by_vid=hbv)
        og = oracle_of(fg, hbv)
        fg.inference(100, 200000, True)
        exact = exact_marginals(og, og.weight["initialValue"].astype(float))
        for v, p in exact.items():
            c0 = int(fg.cstart[v])
            if len(p) == 2:
                got = fg.marginals[c0]
                assert abs(got - p[1]) < 0.01, (v, got, p[1])
            else:
                got = fg.marginals[c0:c0 + len(p)]
                assert np.max(np.abs(got - p)) < 0.01, (v, got, p)


def test_chromatic_and_sequential_agree_statistically():
    """Same stationary distribution from both scan orders: mean magnetisation of a 24x24 grid,
    w=0.2 (well inside the disordered phase), 3000 sweeps each; tolerance 0.01."""
    g = graphgen.ising_grid(24, 24, weight=0.2)
    res = []
    for scan in ("chromatic", "sequential"):
        ns, fg = session(g, seed=42, scan=scan)
        fg.inference(50, 3000 if scan == "chromatic" else 300, True)
        res.append(fg.marginals.mean())
    assert abs(res[0] - 0.5) < 0.01 and abs(res[1] - 0.5) < 0.03


# ------------------------------------------------------------------------------------------
# full-size properties (BASELINE configs #2/#3 shapes)
# ------------------------------------------------------------------------------------------
def test_full_size_grid_properties():
    """1000x1000 grid (config #2): determinism under a seed, tally bounds, symmetry, and agreement
    of one block of rows with the oracle (bit-exact) after several sweeps."""
    g = graphgen.ising_grid(1000, 1000, weight=0.1)
    ns, fg = session(g, seed=20240601)
    info = fg.info()
    assert info["ncolors"] == 2 and info["nowned"] == 1000000 and info["value_bytes"] == 1
    assert abs(info["alg_bytes_inference"] / 1e6 - 106.9) < 0.1
    fg.inference(5, 20, True)
    assert fg.count.min() >= 0 and fg.count.max() <= 20
    assert abs(fg.marginals.mean() - 0.5) < 0.01
    ns2, fg2 = session(g, seed=20240601)
    fg2.inference(5, 20, True)
    assert np.array_equal(fg.count, fg2.count) and np.array_equal(fg.var_value, fg2.var_value)
    ns3, fg3 = session(g, seed=1)
    fg3.inference(5, 20, True)
    assert not np.array_equal(fg.count, fg3.count)
    # oracle on the full grid for 3 sweeps (seconds), bit-exact
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    ns4, fg4 = session(g, seed=8)
    fg4.inference(1, 2, True)
    og.gibbs_dev(order, ps, vv, wv, cnt, 8, 0, True, burnin=True)
    og.gibbs_dev(order, ps, vv, wv, cnt, 8, 1, True)
    og.gibbs_dev(order, ps, vv, wv, cnt, 8, 2, True)
    assert np.array_equal(fg4.var_value[0], vv) and np.array_equal(fg4.count, cnt)


def test_zero_weight_grid_is_uniform():
    g = graphgen.ising_grid(300, 300, weight=0.0)
    ns, fg = session(g, seed=4)
    fg.inference(0, 400, True)
    assert abs(fg.marginals.mean() - 0.5) < 2e-3
    assert abs(fg.marginals.std() - 0.5 / np.sqrt(400)) < 2e-3


def test_learning_recovers_planted_pair_weights():
    """ising.cpp:202-318 scenario (known-answer 2): weights (1, 1, 0.5) planted;
    mini-batch phases of 1000 visits need step*batch < 2, so step 1e-3."""
    g = graphgen.ising_pairs(1000, 1.0, 1.0, 0.5, seed=7)
    ns, fg = session(g, seed=7)
    fg.learn(0, 600, 1e-3, 0.995, 2, 1e-3, 1)
    w = fg.weight_value[0]
    assert abs(w[0] - 1.0) < 0.25 and abs(w[1] - 1.0) < 0.25 and abs(w[2] - 0.5) < 0.25, w


# ------------------------------------------------------------------------------------------
# multi-GPU plumbing, exercised on one GPU
# ------------------------------------------------------------------------------------------
def test_partitioned_sampler_wraps_library_buffers():
    """PartitionedSampler exposes the library's value / weight buffers to torch without copies and
    runs the library on torch's stream."""
    import torch
    from numbskull_amd.distributed import PartitionedSampler
    g = graphgen.ising_grid(64, 64, weight=0.3)
    ns, fg = session(g, seed=5)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    s = PartitionedSampler(fg, None, torch, 0, 1)
    s.gibbs(4)
    torch.cuda.synchronize()
    for k in range(4):
        og.gibbs_dev(order, ps, vv, wv, cnt, 5, k, True)
    assert np.array_equal(s.val.cpu().numpy().astype(np.int64), vv)
    assert s.w.cpu().numpy().tolist() == wv.tolist()
    fg._pull(0, 0)
    assert np.array_equal(fg.count, cnt)


@pytest.mark.parametrize("learn", [False, True])
def test_two_partitions_on_one_gpu_match_emulation(learn):
    """Two handles own the two halves of the variable range (own_range = the reference's shard
    formula).  After every sweep each handle packs its boundary values (nsk_exchange_pack), the
    send buffers are copied into the other handle's gathered buffer -- the all-gather of the
    multi-GPU run -- and unpacked (nsk_exchange_unpack); weights merge as w_start + sum of deltas.
    Owned slices must equal the oracle's emulation of the partitioned semantics bit for bit."""
    import torch
    from numbskull_amd.distributed import PartitionedSampler, shard_range, plan_boundaries
    rng = np.random.default_rng(3)
    if learn:
        g = graphgen.ising_grid(20, 24, weight=0.1, fixed=False, two_weights=True,
                                evidence=rng.integers(0, 2, 480))
    else:
        g = graphgen.ising_grid(20, 24, weight=0.4)
    nvar, world, nsweeps = 480, 2, 5
    parts, oracles = [], []
    for r in range(world):
        ns = numbskull_amd.NumbSkull(quiet=True, seed=31)
        w, v, f, fm, dm, edges = [x.copy() if isinstance(x, np.ndarray) else x for x in g]
        ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=shard_range(r, world, nvar))
        fg = ns.factorGraphs[0]
        ps = PartitionedSampler(fg, None, torch, r, 1)
        ps.world = world                               # drive the exchange by hand below
        parts.append(ps)
        color = fg.colors()
        lo, hi = shard_range(r, world, nvar)
        assert np.all(color[:lo] == -1) and np.all(color[hi:] == -1) and np.all(color[lo:hi] >= 0)
        og = oracle_of(fg)
        oracles.append((og, phases_from_colors(color), og.initial_state(), (lo, hi)))
    needs = [p.fg.ghost_needs() for p in parts]
    assert all(np.array_equal(n, p.fg.ghost_needs(host_only=True)) for n, p in zip(needs, parts))
    lists, slot = plan_boundaries(needs, world, nvar)
    assert slot == 24 and all(len(b) == 24 for b in lists)      # one grid row on each side of the cut
    for p in parts:
        p.install_boundaries(lists, slot)
    L = _lib.lib()
    step = 0.01
    for s in range(nsweeps):
        starts = [p.w.clone() for p in parts]
        ostarts = [st[2].copy() for _, _, st, _ in oracles]
        for p in parts:
            if learn:
                _lib.check(L.nsk_learn_sweeps(p.h, 1, step, 1.0, 2, 0.01, 1, 0))
            else:
                _lib.check(L.nsk_gibbs_sweeps(p.h, 1, 1, 0))
        for og, (order, ps), (vv, ve, wv, cnt), _ in oracles:
            if learn:
                og.learn_call(order, ps, vv, ve, wv, 1, step, 1.0, 2, 0.01, 1, False, 31, s)
            else:
                og.gibbs_dev(order, ps, vv, wv, cnt, 31, s, True)
        step *= 0.9
        for which, sname, rname in ((_lib.BUF_VALUE, "send", "recv"),
                                    (_lib.BUF_VALUE_EVID, "send_evid", "recv_evid")):
            for p in parts:
                _lib.check(L.nsk_exchange_pack(p.h, which))
            torch.cuda.synchronize()
            for q in parts:                                  # the all-gather
                for r, p in enumerate(parts):
                    getattr(q, rname)[r * slot:(r + 1) * slot] = getattr(p, sname)
            torch.cuda.synchronize()
            for p in parts:
                _lib.check(L.nsk_exchange_unpack(p.h, which))
        for r in range(world):                               # oracle side: owners publish their slices
            lo, hi = shard_range(r, world, nvar)
            for q in range(world):
                if q != r:
                    oracles[q][2][0][lo:hi] = oracles[r][2][0][lo:hi]
                    oracles[q][2][1][lo:hi] = oracles[r][2][1][lo:hi]
        if learn:                                            # w = w_start + sum of deltas
            total = sum(p.w - s0 for p, s0 in zip(parts, starts))
            for p, s0 in zip(parts, starts):
                p.w.copy_(s0 + total)
            ototal = sum(st[2] - s0 for (_, _, st, _), s0 in zip(oracles, ostarts))
            for (_, _, st, _), s0 in zip(oracles, ostarts):
                st[2][:] = s0 + ototal
        torch.cuda.synchronize()
    for r in range(world):
        vv, ve, wv, cnt = oracles[r][2]
        lo, hi = shard_range(r, world, nvar)
        got = parts[r].val.cpu().numpy().astype(np.int64)
        assert np.array_equal(got[lo:hi], vv[lo:hi])
        other = lists[1 - r]                                 # the boundary of the other rank arrived
        assert np.array_equal(got[other], vv[other])
        if learn:
            assert np.array_equal(parts[r].val_evid.cpu().numpy().astype(np.int64)[lo:hi], ve[lo:hi])
            assert np.allclose(parts[r].w.cpu().numpy(), wv, rtol=0, atol=1e-15)


@pytest.mark.parametrize("reg", [0, 1, 2])
@pytest.mark.parametrize("shared", [False, True])
def test_weights_with_one_factor_are_updated_in_place(reg, shared):
    """One weight per factor (feature-weighted graphs): the members of a factor lie in different colour
    classes, so such a weight has at most one visit per class and the learning kernels apply its update at
    that visit -- no accumulator, no pass of the update launch over it (nsk_graph_info.direct_weights).  Same
    arithmetic as the per-class rule with K = 1: weights, both chains bit-exact vs the oracle; with `shared`
    a third of the factors are tied to 50 common weights (those keep the accumulators)."""
    g = list(graphgen.boolean_weighted_graph(3000, seed=4))
    rng = np.random.Generator(np.random.PCG64(8))
    w = g[0].copy()
    w["isFixed"] = False
    w["initialValue"] = 0.0
    w["isFixed"][::17] = True                              # a few fixed ones stay out of it
    g[0] = w
    v = g[1].copy()
    v["isEvidence"] = rng.random(len(v)) < 0.5
    v["initialValue"] = rng.integers(0, 2, len(v))
    g[1] = v
    if shared:
        f = g[2].copy()
        pick = rng.random(len(f)) < 0.33
        f["weightId"][pick] = rng.integers(0, 50, int(pick.sum()))
        g[2] = f
    ns, fg = session(tuple(g), seed=6)
    info = fg.info()
    assert info["direct_weights"] > len(w) // 2 and info["learn_lag"] == 0
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 4, 0.02, 0.9, reg, 0.05, 2, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 4, 0.02, 0.9, reg, 0.05, 2, True, 6, 0) == 0
    assert np.array_equal(fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.abs(wv).max() > 0 and (wv[::17] == 0).all()
    fg.inference(0, 3, True)                               # the weights the kernels left are the ones inference reads
    for s in range(4, 7):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 6, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    # the device table keeps these weights in the order the layout meets them (nsk_graph_get_weight_slots):
    # a permutation of the single-factor weights among themselves, invisible through upload / download
    slots = fg.weight_slots()
    assert np.array_equal(np.sort(slots), np.arange(len(w))) and (slots != np.arange(len(w))).any()
    moved = np.nonzero(slots != np.arange(len(w)))[0]
    assert not w["isFixed"][moved].any() and not w["isFixed"][slots[moved]].any()
    nfac_of = np.bincount(fg.factor["weightId"], minlength=len(w))
    assert (nfac_of[moved] == 1).all() and (nfac_of[slots[moved]] == 1).all()
    wv += rng.normal(0, 0.2, len(wv))                      # every weight distinct: the upload must place each one
    fg.weight_value[0][:] = wv
    fg.learn(0, 2, 0.02, 0.9, reg, 0.05, 2, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 2, 0.02, 0.9, reg, 0.05, 2, True, 6, 7) == 0
    assert np.array_equal(fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)


@pytest.mark.parametrize("parts", [1, 5])
def test_shape_classes_per_id_range(parts, monkeypatch):
    """Shape classes are formed per id range of the graph (large graphs: ranges of >= 2^19 ids); forced on a
    small graph here.  Inference and learning stay bit-exact vs the oracle on the resulting layout, and the
    ranges put more variables into general tiles (the leftovers of every shape in every range)."""
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_SHAPE_PARTS", str(parts))
    if parts > 1:       # ... with shape tiles of up to 16 words for every variable (the product keeps lists of
        monkeypatch.setenv("NSK_SHAPE_MAX_WORDS", "16")     # more than 4 words for the entry-parallel groups)
    g = list(graphgen.boolean_weighted_graph(40000, seed=14))
    rng = np.random.Generator(np.random.PCG64(3))
    w = g[0].copy()
    w["isFixed"] = False
    w["initialValue"] = rng.normal(0, 0.2, len(w))
    g[0] = w
    v = g[1].copy()
    v["isEvidence"] = rng.random(len(v)) < 0.5
    v["initialValue"] = rng.integers(0, 2, len(v))
    g[1] = v
    ns, fg = session(tuple(g), seed=12)
    lay, col = fg.layout(), fg.colors()
    if parts > 1:       # a shape tile's 64 lanes come from one fifth of the ids (where whole classes were laid out)
        vids = np.nonzero(col >= 0)[0]
        tile = lay[vids] // 64
        lo = np.full(int(tile.max()) + 1, len(v)); hi = np.full(int(tile.max()) + 1, -1)
        np.minimum.at(lo, tile, vids); np.maximum.at(hi, tile, vids)
        used = hi >= 0
        assert np.median((hi - lo)[used]) <= len(v) // parts
    og = oracle_of(fg)
    order, ps = phases_from_colors(col)
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 3, 0.02, 0.9, 2, 0.05, 1, learn_non_evidence=True)
    assert og.learn_call(order, ps, vv, ve, wv, 3, 0.02, 0.9, 2, 0.05, 1, True, 12, 0) == 0
    assert np.array_equal(fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    fg.inference(1, 3, True)
    assert og.gibbs_dev(order, ps, vv, wv, cnt, 12, 3, True, burnin=True) == 0
    for s_ in range(4, 7):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 12, s_, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_weight_slots_are_the_identity_on_shared_weights():
    """Graphs whose weights are shared by many factors keep the caller's numbering."""
    ns, fg = session(graphgen.mixed_lr_graph(3000, seed=5, nweights=400), seed=2, head_by_vid=True)
    assert fg.info()["direct_weights"] == 0
    assert np.array_equal(fg.weight_slots(), np.arange(400))


def test_table_segments_do_not_read_position_zero():
    """The draw-table kernels take a member's value as its neighbourhood bit.  The ignored slot of a
    member-less entry (ISTRUE) and the padding of a uniform tile used to read "variable 0": with a
    categorical variable at internal id 0 holding 2 they set the NEXT slot's bit -- a wrong table entry and
    wrong satisfied bits (one weight off by 2 * step per epoch on this very graph: the first shard of the
    two-rank LR test).  They read an always-zero id now."""
    from numbskull_amd.distributed import shard_range
    g = graphgen.mixed_lr_graph(4000, seed=12, nweights=300)
    ns = numbskull_amd.NumbSkull(quiet=True, seed=31, head_by_vid=True)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                       own_range=shard_range(0, 2, 4000))
    fg = ns.factorGraphs[0]
    info = fg.info()
    assert info["ztab_entries"] > 0                        # table segments with several slots
    lay = fg.layout()
    v0 = int(np.nonzero(lay == 0)[0][0])                   # the variable at internal id 0 is categorical
    assert fg.variable[v0]["cardinality"] > 2
    og = oracle_of(fg, True)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.learn(0, 5, 0.01, 0.9, 2, 0.01, 1)
    assert og.learn_call(order, ps, vv, ve, wv, 5, 0.01, 0.9, 2, 0.01, 1, False, 31, 0) == 0
    lo, hi = shard_range(0, 2, 4000)
    assert np.array_equal(fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0][lo:hi], vv[lo:hi]) and np.array_equal(fg.var_value_evid[0][lo:hi], ve[lo:hi])
    fg.inference(0, 6, True)
    for s in range(5, 11):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 31, s, True) == 0
    cs = fg.cstart
    assert np.array_equal(fg.var_value[0][lo:hi], vv[lo:hi]) and np.array_equal(fg.count[cs[lo]:cs[hi]], cnt[cs[lo]:cs[hi]])


def test_peer_to_peer_timeout_is_reported_once(monkeypatch):
    """A peer whose flag never arrives: the wait kernel gives up after NSK_P2P_TIMEOUT_S, nsk_p2p_check
    reports it (RuntimeError) and clears the mark, so the handle is usable again; set-up errors are
    refused up front."""
    import ctypes as C
    import torch
    from numbskull_amd.distributed import PartitionedSampler, shard_range
    monkeypatch.setenv("NSK_P2P_TIMEOUT_S", "0.5")
    g = graphgen.ising_grid(16, 16, weight=0.2)
    L = _lib.lib()
    parts = []
    for r in range(2):
        ns = numbskull_amd.NumbSkull(quiet=True, seed=3)
        ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                           own_range=shard_range(r, 2, 256))
        with torch.cuda.stream(torch.cuda.Stream()):
            ps = PartitionedSampler(ns.factorGraphs[0], None, torch, r, 1, nvar_global=256)
        ps.world = 2
        parts.append(ps)
    with pytest.raises(ValueError):                        # nothing set up yet
        _lib.check(L.nsk_p2p_exchange(parts[0].h, 0, 0))
    needs = [p.global_needs() for p in parts]
    bases = (C.c_void_p * 2)()
    for p in parts:
        p.all_needs = needs
        _lib.check(p.p2p_setup())
        b = C.c_void_p()
        _lib.check(L.nsk_p2p_export(p.h, None, C.byref(b)))
        bases[p.rank] = b.value
    for p in parts:
        _lib.check(L.nsk_p2p_import_local(p.h, bases))
        p.p2p = True
    # rank 0 exchanges alone: rank 1 never pushes
    _lib.check(L.nsk_p2p_exchange(parts[0].h, 0, 0))
    with pytest.raises(RuntimeError):
        parts[0].check()
    parts[0].check()                                       # reported once
    # rank 1 catches up (its push carries tag 1), then both exchange properly
    _lib.check(L.nsk_p2p_exchange(parts[1].h, 0, 0))
    parts[1].check()
    for part in (1, 2):
        for p in parts:
            _lib.check(L.nsk_p2p_exchange(p.h, 0, part))
    for p in parts:
        p.check()
    # the payload self-test (pattern pushed, compared on the receiving side) passes on both parities
    for _ in range(2):
        for part in (1, 2, 3):
            for p in parts:
                _lib.check(L.nsk_p2p_selftest(p.h, 1, part))
    for p in parts:
        p.check()
    send = np.zeros(3, np.int32)
    off = np.array([0, 0, 3], np.int64)
    zero = np.zeros(3, np.int64)
    with pytest.raises((ValueError, IndexError)):          # a send list naming variables the handle does not own
        _lib.check(L.nsk_p2p_setup(parts[0].h, 2, 0, _lib.ptr(np.array([200, 201, 202], np.int32)), _lib.ptr(off),
                                   _lib.ptr(send), _lib.ptr(np.array([0, 0, 0], np.int64)), _lib.ptr(zero), _lib.ptr(np.array([0, 3, 0], np.int64))))


@pytest.mark.parametrize("learn", [False, True])
def test_shards_draw_from_disjoint_generator_streams(learn):
    """Generator ids are positions in a handle's own layout, so position q exists in every shard:
    the shard tag in the Philox counter (the first owned variable id, nsk_internal.h rng_tag) must
    keep their uniforms apart.  4096 uncoupled fair coins (ISTRUE, weight 0) cut into two shards:
    with a shared counter the coin at position q of shard 0 would equal the coin at position q of
    shard 1 in every sweep; with disjoint streams they agree half of the time."""
    from test_cabi import _graph_from_spec
    from numbskull_amd.distributed import shard_range
    nvar = 4096
    g = _graph_from_spec(nvar, [(4, [i]) for i in range(nvar)], weights=(0.0,))
    if learn:
        g[0]["isFixed"][:] = True                      # learning sweeps, weights untouched
    vals, pos = [], []
    for r in range(2):
        ns = numbskull_amd.NumbSkull(quiet=True, seed=77)
        ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                           own_range=shard_range(r, 2, nvar))
        fg = ns.factorGraphs[0]
        lo, hi = shard_range(r, 2, nvar)
        rows = []
        for s in range(8):
            if learn:
                fg.learn(0, 1, 0.01, 1.0, 0, 0.0, 1, learn_non_evidence=True)
            else:
                fg.inference(0, 1, True)
            rows.append(fg.var_value[0][lo:hi].copy())
        ids = fg.layout()[lo:hi]
        order = np.argsort(ids)                        # both shards listed by position in their layout
        pos.append(ids[order])
        vals.append(np.array(rows)[:, order])
    assert np.array_equal(pos[0], pos[1])              # same positions (same generator ids) in both shards
    agree = float((vals[0] == vals[1]).mean())
    assert 0.45 < agree < 0.55, agree                  # 32768 pairs of fair coins: sigma = 0.0028
    for v in vals:
        assert 0.45 < float(v.mean()) < 0.55


@pytest.mark.parametrize("reg", [0, 2])
def test_large_feature_values_widen_the_gradient_accumulator(reg):
    """featureValue 1000 on one weight shared by 319 200 factors: the bound on the weight's gradient
    sum in one colour class (|featureValue| x span x arity x factors = 1.3e9) exceeds Q31.32, which
    round 2 refused (NSK_E_RANGE).  The sums now trade fraction bits for range -- Q(31+s).(32-s),
    nsk_graph_info.grad_shift -- stay integer (order-free, deterministic) and the oracle mirrors the
    scale: weights bit-exact.  The reference accumulates in float64 (learning.py:109)."""
    rng = np.random.default_rng(4)
    g = list(graphgen.ising_grid(400, 400, weight=0.0, fixed=False, evidence=rng.integers(0, 2, 160000)))
    fac = g[2].copy()
    fac["featureValue"] = 1000.0
    g[2] = fac
    ns, fg = session(tuple(g), seed=3)
    info = fg.info()
    assert info["grad_shift"] >= 1
    fg.learn(0, 3, 1e-9, 0.9, reg, 0.01, 1)
    og = oracle_of(fg)
    assert og.g.grad_shift == info["grad_shift"]
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    assert og.learn_call(order, ps, vv, ve, wv, 3, 1e-9, 0.9, reg, 0.01, 1, False, 3, 0) == 0
    assert np.array_equal(fg.weight_value[0], wv), (fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0], vv) and wv[0] != 0.0


@pytest.mark.parametrize("burn", [0, 37])
def test_captured_sweep_sequences_equal_the_oracle(burn):
    """A handle whose sweep is table launches only replays 64 or 16 sweeps per hipGraph launch (sweep index
    in device memory + a per-node offset, nsk_gibbs.hip): 64- and 16-sweep replays, the eager remainder, the
    position-tally fold between replays (300 tallied sweeps = 4 x 64 + 2 x 16 + 12) and a burn-in graph (37 = 2 x 16
    + 5) must leave values and tallies exactly where the oracle's sweep-by-sweep run leaves them."""
    g = graphgen.ising_grid(48, 40, weight=0.3)
    ns, fg = session(g, seed=9)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    fg.inference(burn, 300, True)
    for s in range(burn + 300):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 9, s, True, burnin=s < burn) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    fg.inference(0, 21, True)                       # a second call continues the sweep index
    for s in range(burn + 300, burn + 321):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 9, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_captured_sweep_sequences_follow_a_reseed():
    """The captured sequence reads the Philox key and the shard tag from device memory like the sweep
    index: after set_seed the replays draw from the NEW stream (a graph captured under the old seed
    used to keep it while the eager remainder switched)."""
    g = graphgen.ising_grid(48, 40, weight=0.3)
    ns, fg = session(g, seed=9)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    fg.inference(0, 40, True)                       # captures under seed 9
    for s in range(40):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 9, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    fg.set_seed(77, 5)
    fg.inference(0, 40, True)                       # replays + remainder under seed 77 from sweep 5
    for s in range(5, 45):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 77, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    _lib.check(_lib.lib().nsk_set_rng_tag(fg._engine(), 12345))
    og.set_rng_tag(12345)
    fg.inference(0, 40, True)
    for s in range(45, 85):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 77, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


@pytest.mark.parametrize("name", ["grid32", "mixed", "lr_manyw"])
def test_learning_without_the_one_class_lag(golden, name):
    """learn_lag=False (nsk_set_learn_lag 0): every colour class waits for the previous class's weight
    update -- round 3's rule -- and equals the oracle's device mode without a lag array; with the lag
    (the default) the same graph learns different, equally valid weights."""
    g, hbv = _small_graphs(golden)[name]
    if name == "grid32":
        rng = np.random.default_rng(1)
        g = graphgen.ising_grid(32, 32, weight=0.2, fixed=False, two_weights=True, evidence=rng.integers(0, 2, 32 * 32))
    out = {}
    for lag in (False, True):
        ns, fg = session(g, seed=5, head_by_vid=hbv, no_learn_lag=not lag)
        og = oracle_of(fg, hbv)
        order, ps = phases_from_colors(fg.colors())
        vv, ve, wv, _ = og.initial_state()
        fg.learn(0, 4, 0.01, 0.9, 2, 0.05, 1, learn_non_evidence=True)
        assert fg.info()["learn_lag"] == int(lag and len(wv) <= 256)      # (large weight tables never lag)
        assert og.learn_call(order, ps, vv, ve, wv, 4, 0.01, 0.9, 2, 0.05, 1, True, 5, 0, lag=lag and len(wv) <= 256) == 0
        assert np.array_equal(fg.weight_value[0], wv) and np.array_equal(fg.var_value[0], vv)
        assert np.array_equal(fg.var_value_evid[0], ve)
        out[lag] = wv.copy()
    assert name != "grid32" or not np.array_equal(out[False], out[True])


def test_native_rccl_loop_single_rank():
    """nsk_comm_init + nsk_gibbs_sweeps_exchange / nsk_learn_sweeps_exchange with a 1-rank
    communicator: the native loop (sweep, pack, ncclAllGather, unpack, ncclAllReduce of weight
    deltas) must leave a single partition exactly where the plain sweeps leave it."""
    import torch
    from numbskull_amd.distributed import PartitionedSampler
    rng = np.random.default_rng(5)
    g = graphgen.ising_grid(40, 32, weight=0.2, fixed=False, two_weights=True,
                            evidence=rng.integers(0, 2, 1280))
    ns, fg = session(g, seed=17)
    ps = PartitionedSampler(fg, None, torch, 0, 1)
    lists = [np.arange(100, 164, dtype=np.int32)]          # pretend someone reads these
    ps.install_boundaries(lists, 64)
    assert ps._init_native()
    L = _lib.lib()
    _lib.check(L.nsk_gibbs_sweeps_exchange(ps.h, 3, 1, 0))
    _lib.check(L.nsk_learn_sweeps_exchange(ps.h, 2, 1e-3, 0.9, 2, 0.01, 1, 0))
    torch.cuda.synchronize()
    ns2, fg2 = session(g, seed=17)
    fg2.inference(0, 3, True)
    fg2.learn(0, 1, 1e-3, 0.9, 2, 0.01, 1)          # (the exchange loop is one nsk_learn_sweeps call per epoch:
    fg2.learn(0, 1, 1e-3 * 0.9, 0.9, 2, 0.01, 1)    #  the one-class lag pipeline drains at every epoch's merge)
    assert np.array_equal(ps.val.cpu().numpy().astype(np.int64), fg2.var_value[0])
    assert np.array_equal(ps.val_evid.cpu().numpy().astype(np.int64), fg2.var_value_evid[0])
    assert np.allclose(ps.w.cpu().numpy(), fg2.weight_value[0], rtol=0, atol=1e-15)
    assert np.array_equal(ps.send.cpu().numpy(), ps.val.cpu().numpy()[100:164])
    assert np.array_equal(ps.recv.cpu().numpy(), ps.send.cpu().numpy())


def test_rccl_all_gather_on_library_memory():
    """A 1-rank RCCL ("nccl") group runs the exact collective calls of the multi-GPU path
    (in-place all_gather_into_tensor on the int8 value buffer the library allocated, all-reduce of
    the float64 weight delta) -- the 8-GPU run is the driver's, this checks the API contract."""
    import socket
    import torch
    import torch.distributed as dist
    from numbskull_amd.distributed import PartitionedSampler, merge_weight_deltas
    from util import free_port
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g = graphgen.ising_grid(48, 50, weight=0.2, fixed=False)
        ns, fg = session(g, seed=2)
        ps = PartitionedSampler(fg, dist, torch, 0, 1)
        ps.gibbs(3)
        before = ps.val.clone()
        ps.install_boundaries([np.arange(7, 71, dtype=np.int32)], 64)
        ps.world = 1
        ps._exchange(_lib.BUF_VALUE, ps.send, ps.recv)       # pack, all_gather_into_tensor, unpack
        assert torch.equal(ps.recv, ps.val[7:71])
        start = ps.w.clone()
        ps.w += 0.25
        merge_weight_deltas(dist, ps.w, start)
        torch.cuda.synchronize()
        assert torch.equal(before, ps.val)
        assert torch.allclose(ps.w, start + 0.25)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------
def test_edge_case_graphs_run_and_match_oracle():
    """empty graph; isolated variables (uniform draws); cardinality-1 variables; a variable listed
    twice in one factor; a 40-member hub factor (40 colours)"""
    from test_cabi import _graph_from_spec
    ns, fg = session(_graph_from_spec(0, []))
    fg.inference(2, 3, True)
    fg.learn(0, 2, 0.1, 0.9, 2, 0.01, 1)
    assert fg.count.shape == (0,)
    cases = [
        _graph_from_spec(200, []),
        _graph_from_spec(3, [(4, [0]), (3, [1, 1]), (1, [2, 2, 0])], card=np.array([1, 2, 2])),
        _graph_from_spec(40, [(2, list(range(40)))]),
        _graph_from_spec(70, [(3, [i, (i + 1) % 70]) for i in range(70)] + [(-1, [5])],
                         weights=(0.5, -0.3)),
    ]
    for g in cases:
        ns, fg = session(g, seed=13)
        og = oracle_of(fg)
        order, ps = phases_from_colors(fg.colors())
        vv, ve, wv, cnt = og.initial_state()
        fg.inference(1, 6, True)
        og.gibbs_dev(order, ps, vv, wv, cnt, 13, 0, True, burnin=True)
        for s in range(6):
            og.gibbs_dev(order, ps, vv, wv, cnt, 13, 1 + s, True)
        assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
        fg.learn(0, 3, 0.05, 0.9, 1, 0.02, 2, learn_non_evidence=True)
        assert og.learn_call(order, ps, vv, ve, wv, 3, 0.05, 0.9, 1, 0.02, 2, True, 13, 7) == 0
        assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.weight_value[0], wv)
    # isolated variables are fair coins
    ns, fg = session(_graph_from_spec(2000, []), seed=3)
    fg.inference(0, 500, True)
    assert abs(fg.marginals.mean() - 0.5) < 0.005


def test_tally_survives_more_than_255_sweeps_and_repeated_calls():
    """the fast path tallies in uint8 per position and folds every 255 sweeps"""
    g = graphgen.ising_grid(16, 16, weight=0.3)
    ns, fg = session(g, seed=21)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    sweep = 0
    for n in (300, 1, 254, 600):
        fg.inference(0, n, True)
        for _ in range(n):
            og.gibbs_dev(order, ps, vv, wv, cnt, 21, sweep, True)
            sweep += 1
        assert np.array_equal(fg.count, cnt), n
    assert fg.count.max() > 255


# ------------------------------------------------------------------------------------------
# every factor function on the device (the reference's loadfg.py runs each one; here each one is
# also checked against the oracle, which G1 pins to the reference)
# ------------------------------------------------------------------------------------------
def _all_functions_graph():
    from numbskull_amd.inference import FACTORS
    from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar
    rng = np.random.default_rng(23)
    nvar = 14
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = [2, 3, 2, 3, 3, 2, 3, 2, 3, 3, 2, 3, 2, 3]
    variable["isEvidence"] = rng.random(nvar) < 0.4
    variable["initialValue"] = (rng.random(nvar) * variable["cardinality"]).astype(np.int64)
    funcs = sorted(FACTORS.values())
    spec = []
    for fn in funcs:
        for rep in range(2):
            members = rng.choice(nvar, size=3, replace=False).tolist()
            spec.append((fn, members))
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(3 * len(spec), FactorToVar)
    for i, (fn, members) in enumerate(spec):
        factor[i] = (fn, i % 5, [1.0, 0.5, 2.0][i % 3], 3, 3 * i)
        for j, m in enumerate(members):
            fmap[3 * i + j] = (m, int(rng.integers(0, 3)))
    weight = np.zeros(5, Weight)
    weight["initialValue"] = [0.4, -0.3, 0.2, 0.6, -0.5]
    weight["isFixed"] = [False, False, True, False, False]
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), 3 * len(spec)


@pytest.mark.parametrize("scan", ["chromatic", "sequential"])
def test_every_factor_function_on_device(scan):
    g = _all_functions_graph()
    ns, fg = session(g, seed=41, head_by_vid=True, scan=scan)
    assert set(int(f) for f in fg.factor["factorFunction"]) == set(numbskull_amd.inference.FACTORS.values())
    og = oracle_of(fg, True)
    vv, ve, wv, cnt = og.initial_state()
    if scan == "chromatic":
        order, ps = phases_from_colors(fg.colors())
        fg.inference(2, 20, True)
        for s in range(22):
            assert og.gibbs_dev(order, ps, vv, wv, cnt, 41, s, True, burnin=s < 2) == 0
        assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
        fg.learn(0, 10, 0.02, 0.9, 1, 0.05, 2, learn_non_evidence=True)
        assert og.learn_call(order, ps, vv, ve, wv, 10, 0.02, 0.9, 1, 0.05, 2, True, 41, 22) == 0
    else:
        np_rng, py_rng = orc.MT(41, "numpy"), orc.MT(41, "python")
        fg.inference(2, 20, True)
        for s in range(22):
            assert og.gibbs_ref(np_rng, vv, wv, cnt, True, burnin=s < 2) == 0
        assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
        fg.learn(0, 10, 0.02, 0.9, 1, 0.05, 2, learn_non_evidence=True)
        step = 0.02
        for s in range(10):
            assert og.learn_ref(np_rng, py_rng, vv, ve, wv, step, 1, 0.05, 2, True) == 0
            step *= 0.9
    assert np.array_equal(fg.var_value[0], vv)
    assert np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv), (fg.weight_value[0], wv)


def test_marginals_within_1e_3_of_exact_after_burn_in():
    """north_star tolerance: |delta marginal| < 1e-3 after burn-in.  4096 disjoint replicas of a 3x4
    Ising grid (w = 0.5 horizontal, 0.3 vertical) are sampled together for 5000 sweeps after 200 of
    burn-in: 2*10^7 samples per cell position, standard error ~3e-4; the exact marginals come from
    enumerating one replica."""
    from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar
    reps, rows, cols = 4096, 3, 4
    w1, v1, f1, fm1, _, _ = graphgen.ising_grid(rows, cols, weight=0.5, fixed=True, two_weights=True)
    w1["initialValue"] = [0.3, 0.5]
    # bias one corner so that marginals are not all 0.5: ISTRUE on cell 0
    n1, nf1 = rows * cols, len(f1)
    nvar = reps * n1
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2
    factor = np.zeros(reps * (nf1 + 1), Factor)
    fmap = np.zeros(reps * (2 * nf1 + 1), FactorToVar)
    fpr, epr = nf1 + 1, 2 * nf1 + 1
    for name in ("factorFunction", "weightId", "featureValue", "arity"):
        block = np.concatenate([f1[name], [4 if name == "factorFunction" else 2 if name == "weightId"
                                           else 1]])
        factor[name] = np.tile(block, reps)
    factor["ftv_offset"] = np.cumsum(factor["arity"]) - factor["arity"]
    vid1 = np.concatenate([fm1["vid"], [0]])
    fmap["vid"] = (np.tile(vid1, reps) + np.repeat(np.arange(reps) * n1, epr))
    weight = np.zeros(3, Weight)
    weight["initialValue"] = [0.3, 0.5, 0.4]
    weight["isFixed"] = True
    ns, fg = session((weight, variable, factor, fmap, np.zeros(nvar, np.bool_), len(fmap)), seed=2024)
    fg.inference(200, 5000, True)
    got = fg.marginals.reshape(reps, n1).mean(axis=0)
    # exact marginals of one replica
    single = (weight, variable[:n1].copy(), factor[:fpr].copy(), fmap[:epr].copy(),
              np.zeros(n1, np.bool_), epr)
    ns1, fg1 = session(single)
    exact = exact_marginals(oracle_of(fg1), weight["initialValue"].astype(float))
    want = np.array([exact[i][1] for i in range(n1)])
    assert np.max(np.abs(got - want)) < 1e-3, (got, want)
    assert abs(want[0] - 0.5) > 0.05           # the bias makes the check non-trivial


@pytest.mark.parametrize("learn", [False, True])
def test_values_outside_their_domain_take_the_exp_path(learn):
    """A caller may write any int into var_value (the reference computes with whatever is there).
    The draw-table kernels index with the neighbours' low bit, so an upload with a value outside
    [0, cardinality) switches the handle to the exp-per-update kernels until the next regular
    upload; either way the sweep equals the oracle's."""
    rng = np.random.default_rng(9)
    g = graphgen.ising_grid(40, 64, weight=0.25, fixed=not learn, two_weights=learn,
                            evidence=rng.integers(0, 2, 40 * 64) if learn else None)
    ns, fg = session(g, seed=17)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    bad = rng.choice(40 * 64, 50, replace=False)
    for arr in (fg.var_value[0], vv) + ((fg.var_value_evid[0], ve) if learn else ()):
        arr[bad] = 2                                       # not a value of a binary variable
    if learn:
        fg.learn(0, 2, 1e-3, 0.9, 2, 0.01, 1)
        assert og.learn_call(order, ps, vv, ve, wv, 2, 1e-3, 0.9, 2, 0.01, 1, False, 17, 0) == 0
        assert np.array_equal(fg.weight_value[0], wv) and np.array_equal(fg.var_value_evid[0], ve)
    else:
        fg.inference(0, 3, True)
        for s in range(3):
            og.gibbs_dev(order, ps, vv, wv, cnt, 17, s, True)
        assert np.array_equal(fg.count, cnt)
    assert np.array_equal(fg.var_value[0], vv)
    # a regular upload re-enables the tables: one more sweep, still equal
    if not learn:
        fg.inference(0, 1, True)
        og.gibbs_dev(order, ps, vv, wv, cnt, 17, 3, True)
        assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
