#!/usr/bin/env python3
"""Hashes of the compiled layouts of a fixed set of graphs (host-only: nsk_graph_plan with NSK_LAYOUT_HASH=1).
Run before and after a change to the graph compiler that must not move anything: the two outputs are equal."""
import json
import os
import sys

os.environ["NSK_LAYOUT_HASH"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np                                        # noqa: E402
import numbskull_amd                                      # noqa: E402
from numbskull_amd import graphgen                        # noqa: E402


def plan(g, own=None, **kw):
    ns = numbskull_amd.NumbSkull(quiet=True, **kw)
    extra = {} if own is None else {"own_range": own}
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]), **extra)
    color, info = ns.factorGraphs[0].plan()
    return {k: int(v) for k, v in info.items()}


def main():
    out = {}
    bw = graphgen.boolean_weighted_graph(120000, seed=9)
    out["boolw_fixed"] = plan(bw)
    bw[0]["isFixed"] = False
    rng = np.random.Generator(np.random.PCG64(1))
    bw[1]["isEvidence"] = rng.random(len(bw[1])) < 0.5
    out["boolw_learn"] = plan(bw)
    lr = graphgen.mixed_lr_graph(150000, seed=4, nweights=3000)
    out["lr"] = plan(lr, head_by_vid=True)
    out["lr_shard"] = plan(lr, own=(30000, 90000), head_by_vid=True)
    out["lr_bigw"] = plan(graphgen.mixed_lr_graph(100000, seed=5, nweights=600000), head_by_vid=True)
    out["grid"] = plan(graphgen.ising_grid(300, 400, weight=0.1))
    ev = rng.integers(0, 2, 120000)
    out["grid_learn"] = plan(graphgen.ising_grid(300, 400, weight=0.0, fixed=False, two_weights=True, evidence=ev))
    out["grid_shard"] = plan(graphgen.ising_grid(300, 400, weight=0.1), own=(0, 60000))
    print(json.dumps(out, sort_keys=True, indent=1))


if __name__ == "__main__":
    main()
