"""Wide quads of the table segments (DESIGN.md section 3 "wide quads"; nsk_compile.h seg_wide): on a graph whose
exact classes hold long affine runs -- grids with rows of >= 384 cells per colour -- the compiler starts every run on
a multiple of 256 positions and k_gibbs_seg_tabw samples four consecutive positions per lane from dword loads, with
the WIDE generator scheme.  The path `inference.gibbsthread -> draw_sample` (numbskull/inference.py:10-52) must come out
bit for bit as the oracle's device mode computes it, whichever kernel flavour takes a quad."""

import numpy as np
import pytest

from numbskull_amd import graphgen

from util import session, oracle_of, phases_from_colors

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _wide_quads_on_small_graphs(monkeypatch):
    """The library lays out wide quads from 400 000 variables per handle on, and its learning launches take them from 12 000
    quads per launch on; these tests use grids the oracle walks in seconds, so they lower the bounds with the diagnostic
    switches.  The 10M grid of tests/test_config3_gpu.py and the 1M grid of tests/test_hip_parity.py take the path at its
    defaults."""
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_WIDE_MIN", "0")
    monkeypatch.setenv("NSK_WIDE_LEARN_MIN", "0")


def _run_and_compare(fg, og, seed, burn, sweeps, sample_evidence=True, chunks=(None,)):
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    fg.inference(burn, sweeps, sample_evidence)
    for s in range(burn + sweeps):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, seed, s, sample_evidence, burnin=s < burn) == 0
    assert np.array_equal(fg.var_value[0], vv), int((fg.var_value[0] != vv).sum())
    assert np.array_equal(fg.count, cnt)
    return order, ps, vv, wv, cnt


# rows of 499/500, 999/1000 and 768/769 cells per colour (odd widths: the two colours' rows differ in length),
# a grid whose runs are too short to pad (no wide quads: the tile-by-tile kernel), three-slot border rows
@pytest.mark.parametrize("rows,cols,expect_wide", [(40, 1000, True), (24, 2000, True), (33, 1537, True),
                                                   (9, 4099, True), (64, 300, False)])
def test_wide_quads_equal_the_oracle(rows, cols, expect_wide):
    g = graphgen.ising_grid(rows, cols, weight=0.3)
    ns, fg = session(g, seed=5)
    info = fg.info()
    assert (info["wide_quads"] > 0) == expect_wide, info
    if expect_wide:
        assert info["wide_quads"] * 10 >= info["tab_quads"] * 9, info       # the layout pads the rows: nearly all of them
        gen = fg.generators()
        assert ((gen >> 41) & 1).sum() * 10 >= len(gen) * 9                 # ... and their variables draw from the wide scheme
    og = oracle_of(fg)
    _run_and_compare(fg, og, 5, 2, 5)


def test_wide_quads_with_two_weights_and_evidence():
    """Evidence variables that are not sampled (sample_evidence = False) split the classes by evidence flag: runs of
    mixed length, segments of both kinds in one launch, exceptions at both ends of a row."""
    rng = np.random.default_rng(3)
    ev = np.where(np.arange(48 * 1200) % 1200 < 900, 0, 1) * rng.integers(0, 2, 48 * 1200)
    g = graphgen.ising_grid(48, 1200, weight=0.2, fixed=True, two_weights=True, evidence=ev)
    g[1]["isEvidence"] = (np.arange(48 * 1200) % 1200 >= 900).astype(g[1]["isEvidence"].dtype)
    for se in (True, False):
        ns, fg = session(g, seed=21)
        og = oracle_of(fg)
        _run_and_compare(fg, og, 21, 1, 4, sample_evidence=se)


def test_wide_quads_in_captured_sequences_and_tally_folds():
    """16-sweep hipGraph replays of the wide kernel, the eager remainder and the uint8 tally fold (300 sweeps)."""
    g = graphgen.ising_grid(16, 1000, weight=0.25)
    ns, fg = session(g, seed=9)
    assert fg.info()["wide_quads"] > 0
    og = oracle_of(fg)
    order, ps, vv, wv, cnt = _run_and_compare(fg, og, 9, 37, 300)
    fg.inference(0, 21, True)
    for s in range(337, 358):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 9, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_wide_quads_keep_their_generators_on_the_exp_path():
    """A value outside its domain switches the handle to the exp-per-update kernels; the positions of wide quads keep
    the wide generator scheme there (k_gibbs_seg), and the tables come back with the next regular upload."""
    rng = np.random.default_rng(9)
    g = graphgen.ising_grid(20, 1000, weight=0.25)
    ns, fg = session(g, seed=17)
    assert fg.info()["wide_quads"] > 0
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    bad = rng.choice(20 * 1000, 50, replace=False)
    for arr in (fg.var_value[0], vv):
        arr[bad] = 2
    fg.inference(0, 3, True)
    for s in range(3):
        og.gibbs_dev(order, ps, vv, wv, cnt, 17, s, True)
    assert np.array_equal(fg.count, cnt) and np.array_equal(fg.var_value[0], vv)
    fg.inference(0, 2, True)
    for s in range(3, 5):
        og.gibbs_dev(order, ps, vv, wv, cnt, 17, s, True)
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


@pytest.mark.parametrize("switch", ["NSK_NO_WIDE", "NSK_NO_RUN_PAD", "NSK_NO_WIDE_KERNEL", "NSK_NO_TABW_REST"])
def test_wide_quad_switches(monkeypatch, switch):
    """The diagnostic switches that take the wide path out again: no descriptors at all, no run padding (half of a
    1000-column grid's quads stay wide), the tile-by-tile kernel over wide-flagged quads is NOT a valid combination
    (the scheme follows the descriptors), so NSK_NO_WIDE_KERNEL only moves the launch to k_gibbs_seg_tab when no quad
    is wide.  NSK_NO_TABW_REST: the quads that are not wide sampled in line by the waves whose turn they are (what a
    launch with more than NSK_TABW_REST_MAX of them does) instead of by workgroups of their own."""
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv(switch, "1")
    g = graphgen.ising_grid(24, 1000, weight=0.3)
    ns, fg = session(g, seed=5)
    info = fg.info()
    if switch == "NSK_NO_WIDE":
        assert info["wide_quads"] == 0
    og = oracle_of(fg)
    _run_and_compare(fg, og, 5, 1, 3)


def test_learning_with_the_other_quads_in_line(monkeypatch):
    """k_learn_seg_tabw with NSK_NO_TABW_REST (see above): same weights and chains."""
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_NO_TABW_REST", "1")
    rng = np.random.default_rng(2)
    g = graphgen.ising_grid(24, 1000, weight=0.2, fixed=False, two_weights=True, evidence=rng.integers(0, 2, 24 * 1000))
    ns, fg = session(g, seed=5)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    fg.learn(0, 2, 1e-3, 0.9, 2, 0.01, 1)
    assert og.learn_call(order, ps, vv, ve, wv, 2, 1e-3, 0.9, 2, 0.01, 1, False, 5, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv), (fg.weight_value[0], wv)


def test_learning_on_a_padded_layout():
    """The run padding moves every position of the grid; the learning kernels (k_learn_seg_tab) sample tile by tile
    over the same layout: both chains and the weights equal the oracle's."""
    rng = np.random.default_rng(1)
    g = graphgen.ising_grid(24, 1000, weight=0.2, fixed=False, two_weights=True, evidence=rng.integers(0, 2, 24 * 1000))
    ns, fg = session(g, seed=5)
    assert fg.info()["wide_quads"] > 0
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    fg.learn(0, 3, 1e-3, 0.9, 2, 0.01, 1)
    assert og.learn_call(order, ps, vv, ve, wv, 3, 1e-3, 0.9, 2, 0.01, 1, False, 5, 0) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.var_value_evid[0], ve)
    assert np.array_equal(fg.weight_value[0], wv), (fg.weight_value[0], wv)


@pytest.mark.parametrize("sweeps", [3, 140, 300])
def test_packed_tally_equals_the_oracle(sweeps):
    """A handle whose every launch is the wide kernel's keeps the tally inside the value bytes while a call runs (bit 0
    the value, bits 1-7 the count; unpacked every <= 127 sweeps and before the call returns): values and counts as the
    oracle's, across the unpack points, across calls, and with the mode switched off."""
    g = graphgen.ising_grid(16, 1000, weight=0.25)
    ns, fg = session(g, seed=13)
    og = oracle_of(fg)
    order, ps, vv, wv, cnt = _run_and_compare(fg, og, 13, 2, sweeps)
    assert set(np.unique(fg.var_value[0])) <= {0, 1}
    fg.inference(0, 5, True)                       # a second call continues from plain values and the same tallies
    for s in range(2 + sweeps, 7 + sweeps):
        assert og.gibbs_dev(order, ps, vv, wv, cnt, 13, s, True) == 0
    assert np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)


def test_packed_tally_switch(monkeypatch):
    monkeypatch.setenv("NSK_DIAG", "1")
    monkeypatch.setenv("NSK_NO_PACK_TALLY", "1")
    g = graphgen.ising_grid(16, 1000, weight=0.25)
    ns, fg = session(g, seed=13)
    og = oracle_of(fg)
    _run_and_compare(fg, og, 13, 2, 20)


@pytest.mark.parametrize("learn,fused", [(False, False), (False, True), (True, False)])
def test_two_shards_of_a_grid_with_wide_quads_match_emulation(monkeypatch, learn, fused):
    """Shards big enough for wide quads (a 5M-variable half of the 10M grid in a real two-rank run; here a 768 x 1024
    grid in two shards with the bounds lowered): the wide-quad inference kernel on a handle with ghosts, the fused
    exchange's tile-by-tile walk over a layout whose positions draw from the wide scheme, and the wide learning kernel
    on a shard -- through the real peer-to-peer path, bit-exact against the partitioned oracle emulation."""
    import test_config5_shards_gpu as shards
    monkeypatch.setattr(shards, "WORLD", 2)
    probe, _, _ = shards.make_parts("grid", (768, 1024), learn, 1)
    assert all(p.fg.info()["wide_quads"] > 0.9 * p.fg.info()["tab_quads"] > 0 for p in probe)      # the shards ARE laid out in wide quads
    for p in probe:
        p.fg.close()
    out = shards.run_case("grid", (768, 1024), learn, "wide2shards (768x1024 grid, two shards)", nsweeps=3 if not fused else 4,
                          hyper=(1e-4, 0.95, 2, 0.01, 1), fused=fused)
    assert fused or sum(out["ghosts_per_rank"]) == 2 * 1024
