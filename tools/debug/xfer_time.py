"""Times nsk_state_upload / nsk_state_download (FactorGraph._push / _pull) on the 10M grid."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numbskull_amd
from numbskull_amd import graphgen
g = graphgen.ising_grid(2500, 4000, weight=0.1)
ns = numbskull_amd.NumbSkull(quiet=True, seed=1)
ns.loadFactorGraph(*g[:5], int(g[5]))
fg = ns.factorGraphs[0]
fg._engine()
for rep in range(3):
    t0 = time.perf_counter(); fg._push(0, 0); t1 = time.perf_counter(); fg._pull(0, 0); t2 = time.perf_counter()
    print("upload %.1f ms  download %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
fg.inference(0, 3, True)
v = fg.var_value[0].copy()
fg._push(0, 0); fg._pull(0, 0)
assert np.array_equal(v, fg.var_value[0])
print("round trip ok")
