"""Sweep time of the mixed LR graph (scaled-down config #5) against its size: separates the per-colour
latency floor from throughput.  Runs on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import numbskull_amd
from numbskull_amd import graphgen

for n in [int(x) for x in sys.argv[1:]] or [500000, 2000000, 8000000]:
    g = graphgen.mixed_lr_graph(n, seed=20240603)
    ns = numbskull_amd.NumbSkull(quiet=True, head_by_vid=True, seed=1)
    ns.loadFactorGraph(*[x for x in g])
    fg = ns.factorGraphs[0]
    fg.burnIn(3, True)
    t = time.time()
    fg.inference(0, 20, True)            # each call also syncs the state with the host: difference it out
    t1 = time.time()
    fg.inference(0, 220, True)
    dt = ((time.time() - t1) - (t1 - t)) / 200
    info = fg.info()
    print("n=%d colours=%d ms/sweep=%.3f updates/s=%.3e us/colour=%.1f" % (n, info["ncolors"], dt * 1e3, n / dt, dt * 1e6 / info["ncolors"]), flush=True)
    del fg, ns
