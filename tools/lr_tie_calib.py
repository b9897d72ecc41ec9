"""Calibration run for the many-weight chromatic-vs-sequential learning tie (prints statistics)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from util import session, graphgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
g = graphgen.mixed_lr_graph(n, seed=21, nweights=300)
res = {}
for scan in ("sequential", "chromatic"):
    t = time.time()
    ns, fg = session(g, seed=5, scan=scan, head_by_vid=True)
    fg.learn(0, 150, 0.01, 0.95, 2, 0.01, 1)
    res[scan] = fg.weight_value[0].copy()
    print(scan, "%.1fs" % (time.time() - t), "clipped", fg.info()["learn_clipped"])
a, b = res["sequential"], res["chromatic"]
ns, fg = session(g, seed=6, scan="chromatic", head_by_vid=True)
fg.learn(0, 150, 0.01, 0.95, 2, 0.01, 1)
c = fg.weight_value[0].copy()
d = np.abs(a - b)
print("weights: std %.3f range %.3f..%.3f" % (a.std(), a.min(), a.max()))
print("seq vs chr: max %.4f mean %.4f median %.4f p95 %.4f corr %.4f" % (d.max(), d.mean(), np.median(d), np.percentile(d, 95), np.corrcoef(a, b)[0, 1]))
d2 = np.abs(b - c)
print("chr vs chr(other seed): max %.4f mean %.4f median %.4f p95 %.4f corr %.4f" % (d2.max(), d2.mean(), np.median(d2), np.percentile(d2, 95), np.corrcoef(b, c)[0, 1]))
visits = np.bincount(g[2]["weightId"], minlength=300)
print("factors per weight: min %d median %d max %d; worst |d| at weights with %s factors" % (visits.min(), np.median(visits), visits.max(), visits[np.argsort(-d)[:5]]))
