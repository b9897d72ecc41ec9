#!/usr/bin/env python3
"""Diagnostic (GPU box): wide-quad table kernel -- bit-exactness against the oracle on grids with long rows, and
event-timed sweeps of the 1M / 10M grids with and without the wide path / captured sequences."""
import os, sys, time, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import numbskull_amd
from numbskull_amd import graphgen

def parity(n, m, sweeps=6):
    from util import session, oracle_of, phases_from_colors
    g = graphgen.ising_grid(n, m, weight=0.3)
    ns, fg = session(g, seed=5)
    info = fg.info()
    og = oracle_of(fg)
    order, ps = phases_from_colors(fg.colors())
    vv, ve, wv, cnt = og.initial_state()
    fg.inference(2, sweeps - 2, True)
    for s in range(sweeps):
        og.gibbs_dev(order, ps, vv, wv, cnt, 5, s, True, burnin=s < 2)
    ok = np.array_equal(fg.var_value[0], vv) and np.array_equal(fg.count, cnt)
    print("parity %dx%d tab_quads %d wide %d -> %s (value mismatches %d)" % (n, m, info["tab_quads"], info["wide_quads"], "OK" if ok else "DIFFERENT", int((fg.var_value[0] != vv).sum())), flush=True)
    return ok

def timed(n, m, sweeps=200):
    g = graphgen.ising_grid(n, m, weight=0.1)
    ns = numbskull_amd.NumbSkull(quiet=True, seed=3)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]))
    fg = ns.factorGraphs[0]
    info = fg.info()
    fg.burnIn(20, True)
    t = time.time(); fg.inference(0, sweeps, True); dt = time.time() - t
    t = time.time(); fg.inference(0, sweeps, True); dt2 = time.time() - t
    print("timed %dx%d wide %d/%d: %.2f / %.2f us per sweep (host clock, incl. state transfer), mean marginal %.4f" % (
        n, m, info["wide_quads"], info["tab_quads"], dt / sweeps * 1e6, dt2 / sweeps * 1e6, float(np.mean(fg.count) / (2 * sweeps))), flush=True)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "parity"):
        for (n, m) in [(40, 1000), (24, 2000), (64, 1000), (33, 1537)]:
            parity(n, m)
    if what in ("all", "timed"):
        for (n, m) in [(1000, 1000), (2500, 4000)]:
            timed(n, m)
