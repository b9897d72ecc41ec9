#!/usr/bin/env python3
"""Degree-skew micro-benchmark: 200 hub variables, each tied to 5000 leaves by EQUAL factors
(1M leaves with ISTRUE priors).  Compares the wave-per-variable hub kernels with the one-lane
generic kernel:  python tools/hub_bench.py ; NSK_DIAG=1 NSK_NO_HEAVY=1 python tools/hub_bench.py
Round-1 measurement on one MI355X: 0.42 ms/sweep vs 9.9 ms/sweep."""
import sys, time, io, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from contextlib import redirect_stdout
import numbskull_amd
from numbskull_amd.numbskulltypes import *
nhub, per = 200, 5000
nleaf = nhub * per
nvar = nhub + nleaf
variable = np.zeros(nvar, Variable); variable["cardinality"] = 2
nf = nleaf * 2
factor = np.zeros(nf, Factor); factor["featureValue"] = 1.0
fmap = np.zeros(nleaf * 3, FactorToVar)
leaf = np.arange(nleaf) + nhub
hub = np.arange(nleaf) // per
# factor 2i: EQUAL(hub, leaf); factor 2i+1: ISTRUE(leaf)
factor["factorFunction"][0::2] = 3; factor["arity"][0::2] = 2
factor["factorFunction"][1::2] = 4; factor["arity"][1::2] = 1
factor["ftv_offset"] = np.cumsum(factor["arity"]) - factor["arity"]
factor["weightId"][1::2] = 1
vid = np.empty(nleaf * 3, np.int64); vid[0::3] = hub; vid[1::3] = leaf; vid[2::3] = leaf
fmap["vid"] = vid
weight = np.zeros(2, Weight); weight["initialValue"] = [0.05, 0.1]; weight["isFixed"] = True
ns = numbskull_amd.NumbSkull(quiet=True, seed=1)
with redirect_stdout(io.StringIO()):
    ns.loadFactorGraph(weight, variable, factor, fmap, np.zeros(nvar, np.bool_), len(fmap))
fg = ns.factorGraphs[0]
fg.inference(2, 2, True)
t = time.time(); fg.inference(0, 50, True); dt = time.time() - t
print("NO_HEAVY" if os.environ.get("NSK_NO_HEAVY") else "heavy", "ms/sweep (incl. host sync)", 1e3 * fg.inference_epoch_time, fg.info()["ncolors"], fg.marginals[:3])
