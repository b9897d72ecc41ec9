"""Per-group timeline of the entry-parallel inference launch (instrumented build:
tools/build_ablations.sh TIMING, run with NSK_LIB=numbskull_amd/variants/libnsk_TIMING.so)."""
import ctypes as C, io, sys, os
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numbskull_amd
from numbskull_amd import graphgen, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
g = graphgen.mixed_lr_graph(n, seed=20240603)
ns = numbskull_amd.NumbSkull(quiet=True, seed=1, head_by_vid=True)
with redirect_stdout(io.StringIO()):
    ns.loadFactorGraph(*g[:5], int(g[5]))
fg = ns.factorGraphs[0]
L, h = _lib.lib(), fg._engine()
_lib.check(L.nsk_gibbs_sweeps(h, 5, 1, 0))
torch.cuda.synchronize()
buf = np.zeros(4 * 65536, np.uint64)
raw = C.CDLL(_lib.LIB_PATH)
raw.nsk_debug_dump(C.c_void_p(buf.ctypes.data), C.c_int(len(buf)))
b = buf.reshape(-1, 4)
hub = b[60000:64096]
hub = hub[hub[:, 0] > 0]
if len(hub):
    dur = (hub[:, 2] - hub[:, 0]).astype(np.int64)
    ent = (hub[:, 3] >> 32).astype(np.int64)
    ep = (hub[:, 3] & 0xFFFFFFFF) > 1000
    order = np.argsort(-dur)[:8]
    print("hub waves %d (entry-parallel %d): ticks mean %.0f p50 %.0f p90 %.0f max %.0f" %
          (len(hub), ep.sum(), dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
    print("  slowest:", [(int(dur[i]), int(ent[i]), bool(ep[i])) for i in order], "(ticks, list entries, entry-parallel)")
b = b[:32768]
b = b[b[:, 0] > 0]
t0, t1, t4 = b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2].astype(np.int64)
ne = (b[:, 3] & 0xFF).astype(np.int64)
print("groups recorded", len(b), "(one launch's worth of slots); memtime ticks")
for name, x in (("first pass entries", t1 - t0), ("total", t4 - t0)):
    print("%-22s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (name, x.mean(), np.median(x), np.percentile(x, 90), x.max()))
for lo, hi in ((0, 5), (5, 7), (7, 9), (9, 12), (12, 17)):
    m = (ne >= lo) & (ne < hi)
    if m.any():
        print("entries %2d-%2d: %6d groups, total mean %8.0f, first pass %8.0f" % (lo, hi - 1, m.sum(), (t4 - t0)[m].mean(), (t1 - t0)[m].mean()))
