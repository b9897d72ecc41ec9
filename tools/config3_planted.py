#!/usr/bin/env python3
"""BASELINE config #3, end to end on one MI355X:

  1. 2500x4000 Ising grid, weights planted at w* = (0.3 vertical, 0.3 horizontal), 200 inference
     sweeps from the all-zero state (seed 20240602) -> configuration X;
  2. the same grid with two free weights (initial 0), every variable evidence = X;
     50 learning epochs, stepsize 1e-7 (every (variable, factor) visit is one SGD step: ~2*10^7
     visits per weight and colour class), decay 0.95, L2 reg_param 0.01;
  3. report learning variable-updates/s and the recovered weights.

Usage: python tools/config3_planted.py [rows cols]      (prints one JSON line)
"""
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numbskull_amd                                  # noqa: E402
from numbskull_amd import graphgen                    # noqa: E402


def load(g, **kw):
    ns = numbskull_amd.NumbSkull(quiet=True, **kw)
    with redirect_stdout(io.StringIO()):
        ns.loadFactorGraph(*g[:5], int(g[5]))
    return ns.factorGraphs[0]


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2500, 4000)
    nvar = rows * cols
    planted = (0.3, 0.3)
    g = graphgen.ising_grid(rows, cols, weight=planted[0], fixed=True, two_weights=True)
    g[0]["initialValue"] = planted
    fg = load(g, seed=20240602)
    fg.inference(0, 200, True)
    x = fg.var_value[0].copy()
    magnetisation = float(x.mean())
    # nearest-neighbour agreement of the sampled configuration (sufficient statistics of the model)
    grid = x.reshape(rows, cols)
    agree_v = float((grid[1:] == grid[:-1]).mean())
    agree_h = float((grid[:, 1:] == grid[:, :-1]).mean())
    fg.close()

    g2 = graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True, evidence=x)
    fg2 = load(g2, seed=20240603)
    epochs, step, decay = 50, 1e-7 * (10 ** 7 / nvar), 0.95
    trace = []
    t0 = time.time()
    for ep in range(epochs):
        fg2.learn(0, 1, step, decay, 2, 0.01, 1)
        step *= decay
        trace.append([float(v) for v in fg2.weight_value[0]])
    wall = time.time() - t0
    # device time of the learning sweeps themselves: one call of 20 epochs with step 0, state resident
    fg2.learn(0, 20, 0.0, 1.0, 0, 0.0, 1)
    print(json.dumps({
        "config": "BASELINE #3: %dx%d grid, planted w*=%s, 200 sampling sweeps, 50 learning epochs" % (rows, cols, planted),
        "sample_magnetisation": magnetisation, "agreement_vertical": agree_v, "agreement_horizontal": agree_h,
        "stepsize": 1e-7 * (10 ** 7 / nvar), "decay": decay, "regularization": 2, "reg_param": 0.01,
        "recovered_weights": trace[-1], "weights_after_10_epochs": trace[9], "weights_after_25_epochs": trace[24],
        "learning_epoch_ms_including_host_sync": 1e3 * wall / epochs,
        "learning_epoch_ms_device": 1e3 * fg2.learning_epoch_time,
        "learning_updates_per_s_device": nvar / fg2.learning_epoch_time,
    }))


if __name__ == "__main__":
    main()
