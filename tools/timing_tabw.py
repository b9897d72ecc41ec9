#!/usr/bin/env python3
"""Per-wave timeline of the wide-quad table launch (instrumented build: tools/build_variant.sh TIMING -DNSK_ABL_TIMING,
run with NSK_LIB=numbskull_amd/variants/libnsk_TIMING.so): s_memtime at wave entry, after the LDS table landed, after the
first trip and at exit, of the LAST class launch of a short run."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import numbskull_amd
from numbskull_amd import graphgen, _lib
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2500, 4000)
g = graphgen.ising_grid(rows, cols, weight=0.1)
ns = numbskull_amd.NumbSkull(quiet=True, seed=1)
ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]))
fg = ns.factorGraphs[0]
L, h = _lib.lib(), fg._engine()
os.environ["NSK_DIAG"] = "1"; os.environ["NSK_NO_GRAPH"] = "1"
_lib.check(L.nsk_gibbs_sweeps(h, 30, 1, 0))
_lib.check(L.nsk_synchronize(h))
buf = np.zeros(4 * 65536, np.uint64)
raw = C.CDLL(_lib.LIB_PATH)
raw.nsk_debug_dump(C.c_void_p(buf.ctypes.data), C.c_int(len(buf)))
b = buf.reshape(-1, 4)[:16384]
b = b[b[:, 0] > 0]
t0 = b[:, 0].astype(np.int64); t1 = b[:, 1].astype(np.int64); t2 = b[:, 2].astype(np.int64)
t3 = (b[:, 3] & ((1 << 56) - 1)).astype(np.int64); trips = (b[:, 3] >> 56).astype(np.int64)
base = t0.min()
print("%dx%d grid: waves recorded %d, trips per wave min %d max %d; ticks of s_memtime (100 MHz wall clock on gfx950?)" % (rows, cols, len(b), trips.min(), trips.max()))
def st(name, x):
    print("%-34s mean %9.1f  p10 %9.1f  p50 %9.1f  p90 %9.1f  max %9.1f" % (name, x.mean(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
st("wave entry after the first wave's", (t0 - base).astype(float))
w = trips > 0
st("entry -> table landed", (t1 - t0)[w].astype(float))
m = trips > 1
st("table landed -> first trip done", (t2 - t1)[m].astype(float))
st("later trips, each", ((t3 - t2)[m] / np.maximum(trips[m] - 1, 1)).astype(float))
st("entry -> exit", (t3 - t0).astype(float))
print("launch span (first entry -> last exit): %d ticks" % (t3.max() - base))
for k in sorted(set(trips.tolist())):
    mk = trips == k
    print("  waves with %d trips: %5d, entry->exit mean %.1f, exit after first entry mean %.1f max %.1f" % (k, mk.sum(), (t3 - t0)[mk].mean(), (t3 - base)[mk].mean(), (t3 - base)[mk].max()))
