"""Per-tile timeline of one general-tile launch (instrumented build: tools/build_ablations.sh TIMING,
run with NSK_LIB=numbskull_amd/variants/libnsk_TIMING.so)."""
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
    ep = hub[:, 3] > 1000
    print("hub waves recorded %d (entry-parallel %d): duration ticks mean %.0f p50 %.0f p90 %.0f max %.0f; generic-walk mean %.0f, entry-parallel mean %.0f"
          % (len(hub), ep.sum(), dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(),
             dur[~ep].mean() if (~ep).any() else 0, dur[ep].mean() if ep.any() else 0))
b = b[:60000]
b = b[b[:, 0] > 0]
t0, t1, t2 = b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2].astype(np.int64)
ln = (b[:, 3] & 0xFFFFFFFF).astype(np.int64)
print("tiles recorded", len(b), "(last launches of the sweep; memtime ticks)")
walk, draw = t1 - t0, t2 - t1
for name, x in (("walk", walk), ("draw+store", draw), ("total", t2 - t0)):
    print("%-11s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (name, x.mean(), np.median(x), np.percentile(x, 90), x.max()))
for lo, hi in ((0, 16), (16, 32), (32, 48), (48, 64), (64, 96), (96, 256)):
    m = (ln >= lo) & (ln < hi)
    if m.any():
        print("len %3d-%3d: %6d tiles, walk mean %8.0f ticks, per word %6.1f" % (lo, hi, m.sum(), walk[m].mean(), (walk[m] / np.maximum(ln[m], 1)).mean()))

# ---- timeline per launch: tiles of one launch share a time window; launches are separated by gaps
order = np.argsort(t0)
ts, te = t0[order], t2[order]
cuts = [0] + [i for i in range(1, len(ts)) if ts[i] - ts[:i].max() > 20000 and ts[i] > te[:i].max()] + [len(ts)]
print("launch windows (ticks): start-to-end span, tiles, mean concurrency, time with < 1/4 of peak concurrency")
for a, b in zip(cuts[:-1], cuts[1:]):
    if b - a < 100:
        continue
    s, e = ts[a:b], te[a:b]
    span = e.max() - s.min()
    grid = np.linspace(s.min(), e.max(), 400)
    conc = np.array([((s <= g) & (e > g)).sum() for g in grid])
    low = (conc < conc.max() / 4).mean()
    first_end = e.min() - s.min()
    print("  span %7d  tiles %6d  peak conc %5d  mean conc %7.1f  low-occupancy time %4.0f%%  last start at %4.0f%%  first tile done at %4.0f%%"
          % (span, b - a, conc.max(), conc.mean(), 100 * low, 100 * (s.max() - s.min()) / span, 100 * first_end / span))
