"""BASELINE configs[2]/[3] at their stated size -- the 2500x4000 (10M-variable) Ising grid of
ising/ising.cpp:134-199 -- on the GPU against the CPU oracle's device mode on the FULL graph.

Inference: burn-in + tallied sweeps, values and tallies bit-exact (gibbsthread, inference.py:10-33).
Learning: two free weights, every variable evidence, two epochs, weights and both chains bit-exact
(learnthread / sample_and_sgd, learning.py:12-125).  Plus the size-independent properties:
determinism under a seed, tally bounds, symmetry of the mean marginal.
"""

import numpy as np
import pytest

from numbskull_amd import graphgen
from util import session, oracle_of, phases_from_colors

pytestmark = pytest.mark.gpu

ROWS, COLS = 2500, 4000


@pytest.fixture(scope="module")
def grid10m():
    return graphgen.ising_grid(ROWS, COLS, weight=0.1)


def test_config3_inference_bit_exact_vs_oracle(grid10m):
    ns, fg = session(grid10m, seed=20240601)
    info = fg.info()
    assert info["nowned"] == ROWS * COLS and info["ncolors"] == 2 and info["value_bytes"] == 1
    assert info["ztab_entries"] > 0                       # the table-driven segment kernels run
    assert abs(info["alg_bytes_inference"] / 1e7 - 106.97) < 0.01
    assert info["layout_bytes_inference"] < info["alg_bytes_inference"]
    fg.inference(2, 3, True)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, _, wv, cnt = og.initial_state()
    for s in range(5):
        og.gibbs_dev(order, ps, vv, wv, cnt, 20240601, s, True, burnin=s < 2)
    assert np.array_equal(fg.var_value[0], vv), "values differ from the oracle on the 10M grid"
    assert np.array_equal(fg.count, cnt), "tallies differ from the oracle on the 10M grid"
    assert fg.count.min() >= 0 and fg.count.max() <= 3


def test_config3_inference_properties(grid10m):
    ns, fg = session(grid10m, seed=7)
    fg.inference(10, 40, True)
    assert fg.count.min() >= 0 and fg.count.max() <= 40
    assert abs(fg.marginals.mean() - 0.5) < 2e-3          # symmetric model
    x = fg.var_value[0].reshape(ROWS, COLS)
    agree = ((x[1:] == x[:-1]).sum() + (x[:, 1:] == x[:, :-1]).sum()) / float((ROWS - 1) * COLS + ROWS * (COLS - 1))
    # EQUAL with weight 0.1 on +-1 values: a pair alone agrees with probability e^0.1 / (e^0.1 + e^-0.1)
    # = 0.5498; the grid's other couplings push it a little higher
    assert 0.55 < agree < 0.58, agree
    ns2, fg2 = session(grid10m, seed=7)
    fg2.inference(10, 40, True)
    assert np.array_equal(fg.count, fg2.count) and np.array_equal(fg.var_value, fg2.var_value)


def test_config3_learning_bit_exact_vs_oracle():
    rng = np.random.Generator(np.random.PCG64(20240602))
    g = graphgen.ising_grid(ROWS, COLS, weight=0.0, fixed=False, two_weights=True,
                            evidence=rng.integers(0, 2, ROWS * COLS))
    ns, fg = session(g, seed=11)
    fg.learn(0, 2, 1e-7, 0.95, 2, 0.01, 1)
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, _ = og.initial_state()
    assert og.learn_call(order, ps, vv, ve, wv, 2, 1e-7, 0.95, 2, 0.01, 1, False, 11, 0) == 0
    assert np.array_equal(fg.weight_value[0], wv), (fg.weight_value[0], wv)
    assert np.array_equal(fg.var_value[0], vv)
    assert np.array_equal(fg.var_value_evid[0], ve)
    assert np.isfinite(fg.weight_value[0]).all()


def test_config3_planted_weights_are_recovered():
    """Config #3 end to end: the 2500x4000 grid sampled for 200 sweeps at planted
    weights (0.3 vertical, 0.3 horizontal), then learned from that configuration as evidence with the
    config's own hyper-parameters (step 1e-7, decay 0.95, L2 0.01, 50 epochs) -- every (variable,
    factor) visit is one SGD step of sample_and_sgd (learning.py:96-125), 10^7 visits per weight and
    colour class, so the per-class step cap (visits x step = 1 > 0.5) is active.  The weights come back
    within 0.01."""
    import warnings
    g = graphgen.ising_grid(ROWS, COLS, weight=0.3, fixed=True, two_weights=True)
    g[0]["initialValue"] = (0.3, 0.3)
    ns, fg = session(g, seed=20240602)
    fg.inference(0, 200, True)
    x = fg.var_value[0].copy()
    fg.close()
    g2 = graphgen.ising_grid(ROWS, COLS, weight=0.0, fixed=False, two_weights=True, evidence=x)
    ns2, fg2 = session(g2, seed=20240603)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        fg2.learn(0, 50, 1e-7, 0.95, 2, 0.01, 1)
    w = fg2.weight_value[0]
    assert fg2.info()["learn_clipped"] > 0 and any("learn_cap" in str(c.message) for c in caught)
    assert abs(w[0] - 0.3) < 0.01 and abs(w[1] - 0.3) < 0.01, w
