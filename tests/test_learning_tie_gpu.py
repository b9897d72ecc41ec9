"""Chromatic ("device-mode") learning tied to the reference's own per-visit SGD.

scan="sequential" reproduces sample_and_sgd (learning.py:46-125) visit by visit -- bit-exact
against reference traces in test_hip_parity.py.  The production scan updates each weight once per
colour class (batch of k visits, step clipped to learn_cap / k when k * step exceeds the cap).
Both rules have the same fixed point; these tests run the SAME graph with IDENTICAL
hyper-parameters (the reference's CLI defaults: stepsize 0.01, decay 0.95, L2 0.01) under both
scans and require the learned weights to agree, and check the planted-weight recovery.
"""

import numpy as np
import pytest

from numbskull_amd import graphgen
from util import session

pytestmark = pytest.mark.gpu

DEFAULTS = dict(step=0.01, decay=0.95, reg=2, reg_param=0.01)     # numbskull.py:18-149


def planted_grid(rows, cols, w, seed, sweeps=300):
    """An evidence configuration sampled (by the library's own sampler) at planted weights."""
    g = graphgen.ising_grid(rows, cols, weight=0.0, fixed=True, two_weights=True)
    g[0]["initialValue"] = w
    ns, fg = session(g, seed=seed)
    fg.inference(sweeps, 0, True)
    return fg.var_value[0].copy()


def learn(g, scan, epochs, seed=5, **kw):
    ns, fg = session(g, seed=seed, scan=scan, **kw)
    fg.learn(0, epochs, DEFAULTS["step"], DEFAULTS["decay"], DEFAULTS["reg"], DEFAULTS["reg_param"], 1)
    return fg.weight_value[0].copy(), fg


@pytest.mark.parametrize("shape", [(32, 32), (64, 64)])
def test_two_weight_grid_chromatic_matches_sequential_at_reference_defaults(shape):
    rows, cols = shape
    ev = planted_grid(rows, cols, (0.3, 0.15), seed=3)
    g = graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True, evidence=ev)
    w_seq, _ = learn(g, "sequential", 150)
    w_chr, fg = learn(g, "chromatic", 150)
    assert np.isfinite(w_chr).all()
    assert fg.info()["learn_clipped"] > 0           # k * step = 10..41 here: the cap was active
    assert np.abs(w_chr - w_seq).max() <= 0.02, (w_chr, w_seq)


def test_pairs_model_chromatic_matches_sequential():
    g = graphgen.ising_pairs(2000, 1.0, 1.0, 0.5, seed=7)
    w_seq, _ = learn(g, "sequential", 150)
    w_chr, _ = learn(g, "chromatic", 150)
    assert np.abs(w_chr - w_seq).max() <= 0.02, (w_chr, w_seq)


def test_lf_graph_chromatic_matches_sequential():
    g = graphgen.lf_graph(0.0, [1.5, 1.0, 0.5], 2000, seed=3)
    w_seq, _ = learn(g, "sequential", 150)
    w_chr, _ = learn(g, "chromatic", 150)
    assert np.abs(w_chr - w_seq).max() <= 0.03, (w_chr, w_seq)


def test_reference_default_stepsize_recovers_planted_weights_1000x1000():
    """The reference's default -s 0.01 on the 1000x1000 two-weight grid: one colour class visits a
    weight 10^6 times (k * step = 10^4); with the step cap the run converges to the planted
    weights instead of diverging."""
    ev = planted_grid(1000, 1000, (0.3, 0.2), seed=11, sweeps=200)
    g = graphgen.ising_grid(1000, 1000, weight=0.0, fixed=False, two_weights=True, evidence=ev)
    w, fg = learn(g, "chromatic", 60)
    assert np.isfinite(w).all()
    assert abs(w[0] - 0.3) < 0.01 and abs(w[1] - 0.2) < 0.01, w


def test_cap_off_reproduces_plain_batch_rule_and_diverges_where_expected():
    """learn_cap=0 is the unclipped batch rule (what round 1 shipped): same weights as the cap when
    k * step is small, runaway weights at the reference defaults on a shared-weight grid."""
    ev = planted_grid(32, 32, (0.3, 0.15), seed=3)
    g = graphgen.ising_grid(32, 32, weight=0.0, fixed=False, two_weights=True, evidence=ev)
    ns, a = session(g, seed=5, learn_cap=0.0)
    ns, b = session(g, seed=5)
    a.learn(0, 20, 1e-4, 0.95, 2, 0.01, 1)          # k * step = 0.1: below the cap
    b.learn(0, 20, 1e-4, 0.95, 2, 0.01, 1)
    assert np.array_equal(a.weight_value, b.weight_value)
    ns, c = session(g, seed=5, learn_cap=0.0)
    c.learn(0, 30, 0.01, 1.0, 2, 0.01, 1)
    assert np.abs(c.weight_value[0]).max() > 2.0     # oscillates far from (0.3, 0.15)


def test_planted_pair_weights_tight():
    """ising.cpp:202-318 scenario (known-answer 2) with enough pairs for the
    estimate itself to be tight: weights (1, 1, 0.5) planted."""
    g = graphgen.ising_pairs(40000, 1.0, 1.0, 0.5, seed=7)
    ns, fg = session(g, seed=7)
    fg.learn(0, 300, 0.01, 0.98, 2, 1e-4, 1)
    w = fg.weight_value[0]
    assert abs(w[0] - 1.0) < 0.05 and abs(w[1] - 1.0) < 0.05 and abs(w[2] - 0.5) < 0.05, w


def test_many_weight_lr_graph_chromatic_matches_sequential():
    """A 300-weight mixed LR graph (40 000 variables, 177..6929 factors per weight, categorical and
    boolean variables, OR / IMPLY_MLN / *_CAT factors -- the shape of config #5) at the reference's
    CLI defaults: the weights the chromatic scan learns must agree with the reference's own per-visit
    trajectory (scan="sequential") as closely as two chromatic runs with different seeds agree with
    each other -- the difference is SGD noise, not a different fixed point.  (Measured: sequential vs
    chromatic mean |dw| 0.018, max 0.083, correlation 0.981; chromatic vs chromatic with another seed
    0.019 / 0.082 / 0.981.)"""
    g = graphgen.mixed_lr_graph(40000, seed=21, nweights=300)
    w_seq, _ = learn(g, "sequential", 100, head_by_vid=True)
    w_chr, fg = learn(g, "chromatic", 100, head_by_vid=True)
    w_chr2, _ = learn(g, "chromatic", 100, seed=6, head_by_vid=True)
    assert fg.info()["learn_clipped"] > 0               # heavy weights: the per-class cap was active
    d, d2 = np.abs(w_seq - w_chr), np.abs(w_chr - w_chr2)
    assert np.corrcoef(w_seq, w_chr)[0, 1] > 0.95, np.corrcoef(w_seq, w_chr)[0, 1]
    assert d.mean() < 0.03 and d.max() < 0.15, (d.mean(), d.max())
    assert d.mean() < 1.5 * d2.mean() + 0.005, (d.mean(), d2.mean())


# ------------------------------------------------------------------------------------------------------------------
# The tie in the north star's own unit: marginals.  Both scans climb the same (regularised) likelihood; what the
# chromatic scan's per-class batches, step cap and one-class lag change is the path and the SGD noise, not the fixed
# point.  Learn with the reference's own trajectory (scan="sequential" = sample_and_sgd, learning.py:46-125) and with
# the production scan at the reference's CLI defaults, then compute the EXACT marginals of one unit of the model
# (a pair, a labelled data point, the grid itself: enumeration, tests/util.py exact_marginals) under both weight
# vectors: max |delta marginal| < 1e-3 once the decayed step has frozen both runs.
def _unit_marginals(unit, w):
    from oracle import binding as orc
    from util import exact_marginals
    import numbskull_amd
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in unit[:5]], int(unit[5]))
    fg = ns.factorGraphs[0]
    og = orc.Graph(fg.weight, fg.variable, fg.factor, fg.fmap, fg.vmap, fg.factor_index)
    m = exact_marginals(og, np.asarray(w, np.float64))
    return np.concatenate([m[i] for i in sorted(m)])


def _tie(g, unit, epochs, **kw):
    w_seq, _ = learn(g, "sequential", epochs, **kw)
    w_chr, _ = learn(g, "chromatic", epochs, **kw)
    d = np.abs(_unit_marginals(unit, w_seq) - _unit_marginals(unit, w_chr)).max()
    print("tie: |dw| max %.5f, max |delta marginal| %.2e  (weights %s / %s)" % (np.abs(w_seq - w_chr).max(), d, w_seq, w_chr))
    return d


def test_marginals_under_both_scans_pairs():
    g = graphgen.ising_pairs(8000, 1.0, 1.0, 0.5, seed=7)
    unit = list(graphgen.ising_pairs(1, 1.0, 1.0, 0.5, seed=7))
    unit[1] = unit[1].copy(); unit[1]["isEvidence"] = 0                    # the pair as a query: both variables free
    assert _tie(g, tuple(unit), 200) < 1e-3


def test_marginals_under_both_scans_lf():
    g = graphgen.lf_graph(0.0, [1.5, 1.0, 0.5], 16000, seed=3)
    unit = list(graphgen.lf_graph(0.0, [1.5, 1.0, 0.5], 1, seed=3))
    unit[1] = unit[1].copy(); unit[1]["isEvidence"] = 0
    assert _tie(g, tuple(unit), 200) < 1e-3


def test_marginals_under_both_scans_small_grid():
    """3 x 4 grid, two weights; the evidence: 2048 independent configurations of that grid sampled at planted weights
    (0.3, 0.15) -- one 12-variable configuration alone has no maximum-likelihood weights worth comparing -- as 2048
    disjoint copies that share the two weights; the marginals: those of the 12-variable grid itself."""
    rows, cols, reps = 3, 4, 2048
    g1 = graphgen.ising_grid(rows, cols, weight=0.0, fixed=True, two_weights=True)
    g1[0]["initialValue"] = (0.3, 0.15)
    ns, fg = session(g1, seed=11)
    cfgs = []
    for _ in range(reps):
        fg.inference(0, 20, True)
        cfgs.append(fg.var_value[0].copy())
    one = graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True, evidence=cfgs[0])
    g = graphgen.replicate(one, cfgs)
    unit = graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True)
    assert _tie(g, unit, 200) < 1e-3
