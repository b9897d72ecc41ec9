"""Pins the CPU oracle (oracle/nsk_oracle.c) to the REFERENCE: every fixture under
tests/golden/ was produced by running HazyResearch/numbskull itself (tools/make_goldens.py).
Bit-exact on integers and on weights (same operation order; libm exp in reference mode)."""

import numpy as np
import pytest

from conftest import graph_from
from oracle import binding as orc


def oracle_graph(g, head_by_vid=False):
    w, v, f, fm, dm, _ = g
    v2, vmap, fi, rc = orc.compute_var_map(v, f, fm, dm)
    assert rc == 0
    return orc.Graph(w, v2, f, fm, vmap, fi, head_by_vid=head_by_vid)


# ---------------------------------------------------------------- G7: generators
@pytest.mark.parametrize("seed", [0, 1, 42, 1234, 20240601, 2 ** 32 - 1])
def test_mt19937_streams(golden, seed):
    z = golden("g7_rng.npz")
    a = orc.MT(seed, "numpy")
    b = orc.MT(seed, "python")
    assert np.array_equal(np.array([a.random() for _ in range(700)]), z["np_%d" % seed])
    assert np.array_equal(np.array([b.random() for _ in range(700)]), z["py_%d" % seed])


def test_mt19937_matches_numpy_live():
    rs = np.random.RandomState(987654321)
    a = orc.MT(987654321, "numpy")
    assert [a.random() for _ in range(2000)] == list(rs.random_sample(2000))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert orc.philox(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert orc.philox(f, f, f, f, f, f) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_exp_det_accuracy():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-40, 40, 20000), rng.uniform(-745, 709, 5000),
                        rng.normal(0, 1, 20000), [0.0, -0.0, 1.0, -1.0, 709.7, -745.0, 1e-300]])
    got = orc.exp_det(x)
    want = np.exp(x)
    ulp = np.spacing(want)
    assert np.all(np.abs(got - want) <= ulp), np.max(np.abs(got - want) / ulp)
    assert orc.exp_det(np.array([800.0]))[0] == np.inf
    assert orc.exp_det(np.array([-800.0]))[0] == 0.0
    assert np.isnan(orc.exp_det(np.array([np.nan]))[0])


# ---------------------------------------------------------------- G1: eval_factor
def test_eval_factor_table(golden):
    z = golden("g1_eval_factor.npz")
    variable, factor, fmap, states, cases = (z[k] for k in
                                             ("variable", "factor", "fmap", "states", "cases"))
    vmap = np.zeros(0, orc.Graph.__init__.__globals__["np"].dtype(
        [("value", "i8"), ("factor_index_offset", "i8"), ("factor_index_length", "i8")]))
    w = np.zeros(1, np.dtype([("isFixed", np.bool_), ("initialValue", "f8")]))
    g = orc.Graph(w, variable, factor, fmap, vmap, np.zeros(0, np.int64))
    status_of = {0: orc.OK, 1: orc.E_FACTOR_FUNC, 2: orc.E_INDEX}
    nbad = 0
    for fid, s, var_samp, value, status, want in cases:
        vv = np.ascontiguousarray(states[int(s)], np.int64)
        rc, got = g.eval_factor(int(fid), int(var_samp), int(value), vv)
        assert rc == status_of[int(status)], (fid, s, var_samp, value, rc, status)
        if rc == 0 and got != want:
            nbad += 1
    assert nbad == 0
    assert len(cases) > 40000
    # the fixture really exercises the literal head-index quirk: intended lookup differs
    g2 = orc.Graph(w, variable, factor, fmap, vmap, np.zeros(0, np.int64), head_by_vid=True)
    ndiff = 0
    for fid, s, var_samp, value, status, want in cases:
        if int(factor[int(fid)]["factorFunction"]) in (13, 16, 17) and status == 0:
            vv = np.ascontiguousarray(states[int(s)], np.int64)
            rc, got = g2.eval_factor(int(fid), int(var_samp), int(value), vv)
            ndiff += int(got != want)
    assert ndiff > 0


# ---------------------------------------------------------------- G2: index build
@pytest.mark.parametrize("tag", ["grid4x5", "mixed", "skiplast", "pairs", "lf"])
def test_compute_var_map(golden, tag):
    z = golden("g2_index_build.npz")
    w, v, f, fm, dm, _ = graph_from(z, tag)
    skip = z[tag + "_in_factors_to_skip"] if tag + "_in_factors_to_skip" in z.files else None
    v2, vmap, fi, rc = orc.compute_var_map(v, f, fm, dm, skip)
    assert rc == 0
    assert np.array_equal(v2, z[tag + "_out_variable"])
    assert np.array_equal(vmap, z[tag + "_out_vmap"])
    want_fi = z[tag + "_out_factor_index"]
    # entries beyond each slot's deduped length are leftovers in both; compare live entries
    for slot in vmap:
        o, n = int(slot["factor_index_offset"]), int(slot["factor_index_length"])
        assert np.array_equal(fi[o:o + n], want_fi[o:o + n])
    assert np.array_equal(fi, want_fi)


# ---------------------------------------------------------------- G3: inference traces
G3_TAGS = ["grid4x5_w05", "grid32_w01", "grid32_w05", "mixed", "mixed_noev", "lf", "headquirk"]


@pytest.mark.parametrize("tag", G3_TAGS)
def test_inference_trace(golden, tag):
    z = golden("g3_inference.npz")
    g = oracle_graph(graph_from(z, tag))
    seed, burn, se = int(z[tag + "_seed"]), int(z[tag + "_burn"]), bool(z[tag + "_sample_evidence"])
    vals, counts = z[tag + "_var_value"], z[tag + "_count"]
    vv, _, wv, cnt = g.initial_state()
    rng = orc.MT(seed, "numpy")
    for _ in range(burn):
        assert g.gibbs_ref(rng, vv, wv, cnt, se, burnin=True) == 0
    assert np.array_equal(vv, vals[0])
    assert not cnt.any()
    for e in range(len(counts)):
        assert g.gibbs_ref(rng, vv, wv, cnt, se, burnin=False) == 0
        assert np.array_equal(vv, vals[e + 1]), (tag, e)
        assert np.array_equal(cnt, counts[e]), (tag, e)


def test_inference_long_run_baseline_md(golden):
    """4x5 grid, w=0.5, burn-in 10, 1000 epochs, seed 42."""
    z = golden("g3_inference.npz")
    g = oracle_graph(graph_from(z, "grid4x5_w05"))
    vv, _, wv, cnt = g.initial_state()
    rng = orc.MT(42, "numpy")
    for _ in range(10):
        g.gibbs_ref(rng, vv, wv, cnt, True, burnin=True)
    for _ in range(1000):
        g.gibbs_ref(rng, vv, wv, cnt, True)
    assert np.array_equal(cnt, z["grid4x5_long_count"])
    assert cnt.tolist()[:5] == [565, 567, 582, 592, 554]
    assert np.array_equal(cnt / 1000.0, z["grid4x5_long_marginals"])


# ---------------------------------------------------------------- G4: learning traces
def _g4_cases():
    for tag in ("pairs", "mixed", "lf"):
        for reg in (0, 1, 2):
            for lne in (0, 1):
                for k in ((1, 3) if reg == 1 else (1,)):
                    yield tag, reg, lne, k


@pytest.mark.parametrize("tag,reg,lne,trunc", list(_g4_cases()))
def test_learning_trace(golden, tag, reg, lne, trunc):
    z = golden("g4_learning.npz")
    g = oracle_graph(graph_from(z, tag))
    name = "%s_r%d_l%d_k%d" % (tag, reg, lne, trunc)
    seed = int(z[name + "_seed"])
    ws, vvs, ves = z[name + "_weights"], z[name + "_var_value"], z[name + "_var_value_evid"]
    vv, ve, wv, _ = g.initial_state()
    np_rng, py_rng = orc.MT(seed, "numpy"), orc.MT(seed, "python")
    step = 0.05
    assert np.array_equal(wv, ws[0])
    for e in range(len(ws) - 1):
        assert g.learn_ref(np_rng, py_rng, vv, ve, wv, step, reg, 0.02, trunc, bool(lne)) == 0
        step *= 0.9
        assert np.array_equal(vv, vvs[e + 1]), (name, e)
        assert np.array_equal(ve, ves[e + 1]), (name, e)
        assert np.array_equal(wv, ws[e + 1]), (name, e, wv, ws[e + 1])


# ---------------------------------------------------------------- device mode: thread-count independence
@pytest.mark.parametrize("reg", [0, 1, 2])
def test_device_mode_is_thread_count_independent(reg):
    """The device-mode sweeps walk a colour class with several host threads (orc_set_threads; the class's variables do
    not read each other, the gradient sums are integers): values, tallies and weights must not depend on their number."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from numbskull_amd import graphgen
    rng = np.random.default_rng(2)
    g = graphgen.ising_grid(120, 130, weight=0.2, fixed=False, two_weights=True, evidence=rng.integers(0, 2, 120 * 130))
    g[1]["isEvidence"] = (rng.random(120 * 130) < 0.5).astype(g[1]["isEvidence"].dtype)
    import numbskull_amd                      # (its host-side index build: plain numpy + C++, no GPU)
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]))
    fg = ns.factorGraphs[0]
    og = orc.Graph(fg.weight, fg.variable, fg.factor, fg.fmap, fg.vmap, fg.factor_index)
    i, j = np.divmod(np.arange(120 * 130), 130)
    color = ((i + j) & 1).astype(np.int32)
    assert og.check_coloring(color) == (-1, -1)
    order = np.concatenate([np.flatnonzero(color == 0), np.flatnonzero(color == 1)]).astype(np.int64)
    ps = np.array([0, int((color == 0).sum()), len(color)], np.int64)
    out = []
    for threads in (1, 7):
        orc.set_threads(threads)
        vv, ve, wv, cnt = og.initial_state()
        for s in range(3):
            assert og.gibbs_dev(order, ps, vv, wv, cnt, 5, s, True) == 0
        assert og.learn_call(order, ps, vv, ve, wv, 3, 0.01, 0.9, reg, 0.05, 2, True, 5, 3) == 0
        out.append((vv.copy(), ve.copy(), wv.copy(), cnt.copy()))
    orc.set_threads(min(64, os.cpu_count() or 1))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.any(out[0][3] != 0) and (reg == 1 or np.any(out[0][2] != 0.0))


# ---------------------------------------------------------------- f3: why UFO aggregation is not built
def test_ufo_returns_its_first_member_on_binary_variables(golden):
    """SURVEY.md section 8 f3 / salt/src/messages.py:1049-1080: the reference's UFO scheme ships, per boundary variable, the
    summed factor-value differences of its remote factors and lets a two-member UFO factor (30) deliver them from a
    helper variable.  eval_factor's UFO branch (inference.py:398-405) returns 0 for x_0 = 0 and member x_0 - 1 otherwise:
    for x_0 = 1 that member is x_0 ITSELF, so on a binary variable -- every variable of BASELINE's partitioned configs --
    the factor's value is x_0 whatever the helper holds: the aggregate cannot reach the potential.  Pinned here on the
    REFERENCE's own table (tests/golden/g1_eval_factor.npz, generated by importing it) and on the oracle; DESIGN.md
    section 5 closes the row with this test's name (partial factors, which do deliver their aggregates, are built)."""
    z = golden("g1_eval_factor.npz")
    variable, factor, fmap, states, cases = (z[k] for k in ("variable", "factor", "fmap", "states", "cases"))
    n_seen = 0
    for fid, s, var_samp, value, status, want in cases:
        fa = factor[int(fid)]
        if int(fa["factorFunction"]) != 30 or int(status) != 0:
            continue
        x0_vid = int(fmap[int(fa["ftv_offset"])]["vid"])
        x0 = int(value) if x0_vid == int(var_samp) else int(states[int(s)][x0_vid])
        if x0 in (0, 1):
            assert int(want) == x0, (fid, s, var_samp, value, want)       # never the helper's value
            n_seen += 1
    assert n_seen >= 100
