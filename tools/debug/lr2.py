"""Debug: the 4000-variable LR graph of tests/multirank_worker.py as 2 range shards in ONE process (whole-graph
handles with own_range, exchange by hand), learning, compared with the oracle emulation sweep by sweep."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from util import graphgen, oracle_of, phases_from_colors
import numbskull_amd
from numbskull_amd import _lib
from numbskull_amd.distributed import shard_range

lag = int(os.environ.get("LAG", "1"))
world = int(os.environ.get("WORLD", "2"))
g = graphgen.mixed_lr_graph(4000, seed=12, nweights=300)
nvar = len(g[1])
if os.environ.get("REMAP"):        # every factor of weight 131 gets a weight of its own: which one goes wrong?
    from numbskull_amd.numbskulltypes import Weight
    idx = np.nonzero(g[2]["weightId"] == 131)[0]
    fac = g[2].copy()
    fac["weightId"][idx] = 300 + np.arange(len(idx))
    g = (np.zeros(300 + len(idx), Weight), g[1], fac) + tuple(g[3:])
    print("remapped factors", list(idx))
L = _lib.lib()
hs, ogs = [], []
for r in range(world):
    ns = numbskull_amd.NumbSkull(quiet=True, seed=31, head_by_vid=True, no_learn_lag=not lag)
    w, v, f, fm, dm, edges = [x.copy() if isinstance(x, np.ndarray) else x for x in g]
    ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=shard_range(r, world, nvar))
    fg = ns.factorGraphs[0]
    og = oracle_of(fg, True)
    hs.append(fg)
    ogs.append((og, phases_from_colors(fg.colors()), og.initial_state()))
    print("rank", r, fg.info())
    if r == 0:
        lay, col = fg.layout(), fg.colors()
        inv = {int(lay[v]): v for v in range(nvar)}
        import json
        json.dump({"inv": {str(k): int(v) for k, v in inv.items()}, "col": [int(x) for x in col],
                   "ev": [int(x) for x in g[1]["isEvidence"]]}, open("gpurun_out/lr2_layout.json", "w"))
step = 0.01
for s in range(4):
    starts = [st[2].copy() for _, _, st in ogs]
    for r, fg in enumerate(hs):
        fg.learn(0, 1, step, 1.0, 2, 0.01, 1)        # push state, one epoch, pull state
    for og, (order, ps), (vv, ve, wv, cnt) in ogs:
        assert og.learn_call(order, ps, vv, ve, wv, 1, step, 1.0, 2, 0.01, 1, False, 31, s, lag=bool(lag)) == 0
    for r, fg in enumerate(hs):
        vv, ve, wv, cnt = ogs[r][2]
        lo, hi = shard_range(r, world, nvar)
        dv = np.nonzero(fg.var_value[0][lo:hi] != vv[lo:hi])[0] + lo
        de = np.nonzero(fg.var_value_evid[0][lo:hi] != ve[lo:hi])[0] + lo
        dw = np.nonzero(fg.weight_value[0] != wv)[0]
        print("sweep", s, "rank", r, "free-chain diffs", len(dv), dv[:8], "evid diffs", len(de), de[:8], "weight diffs", len(dw), dw[:8],
              (fg.weight_value[0][dw[:3]], wv[dw[:3]]) if len(dw) else "")
        if len(de):
            col = fg.colors()
            for v_ in de[:4]:
                print("   var", v_, g[1][v_], "color", col[v_], "nentries", int((g[3]["vid"] == v_).sum()))
    # exchange: owners publish, weights merge (both sides)
    for r in range(world):
        lo, hi = shard_range(r, world, nvar)
        for q in range(world):
            if q != r:
                hs[q].var_value[0][lo:hi] = hs[r].var_value[0][lo:hi]
                hs[q].var_value_evid[0][lo:hi] = hs[r].var_value_evid[0][lo:hi]
                ogs[q][2][0][lo:hi] = ogs[r][2][0][lo:hi]
                ogs[q][2][1][lo:hi] = ogs[r][2][1][lo:hi]
    tot = sum(fg.weight_value[0] - s0 for fg, s0 in zip(hs, starts))
    otot = sum(st[2] - s0 for (_, _, st), s0 in zip(ogs, starts))
    for fg, s0 in zip(hs, starts):
        fg.weight_value[0][:] = s0 + tot
    for (_, _, st), s0 in zip(ogs, starts):
        st[2][:] = s0 + otot
    step *= 0.9
