"""Times tallied vs burn-in sweeps of the 10M grid (the burn-in launch skips the tally load/store:
prices those two vector-memory instructions per tile)."""
import ctypes as C, io, sys, time, os
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numbskull_amd
from numbskull_amd import graphgen, _lib
g = graphgen.ising_grid(2500, 4000, weight=0.1, fixed=True)
ns = numbskull_amd.NumbSkull(quiet=True, seed=1)
with redirect_stdout(io.StringIO()):
    ns.loadFactorGraph(*g[:5], int(g[5]))
fg = ns.factorGraphs[0]
L, h = _lib.lib(), fg._engine()
for burn in (0, 1, 0, 1):
    _lib.check(L.nsk_gibbs_sweeps(h, 20, 1, burn))
    torch.cuda.synchronize()
    _lib.check(L.nsk_profile_begin(h))
    _lib.check(L.nsk_gibbs_sweeps(h, 100, 1, burn))
    ms, nl = C.c_double(), C.c_int64()
    _lib.check(L.nsk_profile_end(h, C.byref(ms), C.byref(nl)))
    print("burnin=%d  %.2f us per launch" % (burn, ms.value * 1e3 / nl.value))
