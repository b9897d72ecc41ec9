"""BASELINE configs[3] ("config #4"): the 2500x4000 (10M-variable) Ising grid range-partitioned into
EIGHT shards with the reference's shard formula (inference.py:17-18), every shard a handle of its
own -- here all on one device, the boundary exchange done by hand exactly as the N-rank loop does it
(nsk_exchange_pack -> the all-gather -> nsk_exchange_unpack; numbskull_master.py:165-224 semantics:
ghost values are one sweep old).  Owned slices must equal the oracle's emulation of the partitioned
run bit for bit, at full size, for inference and for learning (weights merged as w_start + sum of
deltas, numbskull_master.py:223-224).  The per-shard phase timings (HIP events) go to
gpurun_out/config4_shards.json (copied to profiles/ by tools/collect_profiles.sh)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from numbskull_amd import _lib, graphgen
from util import oracle_of, phases_from_colors
import numbskull_amd

pytestmark = pytest.mark.gpu

ROWS, COLS, WORLD = 2500, 4000, 8


def _timed(L, h, fn):
    ms, nl = C.c_double(), C.c_int64()
    _lib.check(L.nsk_profile_begin(h))
    fn()
    _lib.check(L.nsk_profile_end(h, C.byref(ms), C.byref(nl)))
    return ms.value * 1e3


@pytest.mark.parametrize("learn", [False, True])
def test_config4_eight_shards_match_emulation(learn):
    import torch
    from numbskull_amd.distributed import PartitionedSampler, shard_range, plan_boundaries
    rng = np.random.Generator(np.random.PCG64(20240602))
    if learn:
        g = graphgen.ising_grid(ROWS, COLS, weight=0.0, fixed=False, two_weights=True,
                                evidence=rng.integers(0, 2, ROWS * COLS))
    else:
        g = graphgen.ising_grid(ROWS, COLS, weight=0.1)
    nvar, nsweeps, seed = ROWS * COLS, 3, 20240601
    parts, oracles = [], []
    for r in range(WORLD):
        ns = numbskull_amd.NumbSkull(quiet=True, seed=seed)
        ns.loadFactorGraph(g[0].copy(), g[1], g[2], g[3], g[4], int(g[5]), own_range=shard_range(r, WORLD, nvar))
        fg = ns.factorGraphs[0]
        ps = PartitionedSampler(fg, None, torch, r, 1)
        ps.world = WORLD                                    # the exchange is driven by hand below
        parts.append(ps)
        og = oracle_of(fg)                                  # checks the layout and the colouring of every shard
        oracles.append((og, phases_from_colors(fg.colors()), og.initial_state()))
        info = fg.info()
        assert info["nowned"] == shard_range(r, WORLD, nvar)[1] - shard_range(r, WORLD, nvar)[0]
    lists, slot = plan_boundaries([p.fg.ghost_needs() for p in parts], WORLD, nvar)
    # a shard is a band of 312 or 313 grid rows: its neighbours read one row of 4000 values on each side
    assert slot == 2 * COLS and [len(b) for b in lists] == [COLS] + [2 * COLS] * (WORLD - 2) + [COLS]
    for p in parts:
        p.install_boundaries(lists, slot)
    L = _lib.lib()
    timing = {"sweep_us": [], "pack_us": [], "unpack_us": []}
    step = 1e-7
    for s in range(nsweeps):
        starts = [p.w.clone() for p in parts]
        ostarts = [st[2].copy() for _, _, st in oracles]
        t_sweep = []
        for p in parts:
            if learn:
                t_sweep.append(_timed(L, p.h, lambda: _lib.check(L.nsk_learn_sweeps(p.h, 1, step, 1.0, 2, 0.01, 1, 0))))
            else:
                t_sweep.append(_timed(L, p.h, lambda: _lib.check(L.nsk_gibbs_sweeps(p.h, 1, 1, 0))))
        timing["sweep_us"].append(t_sweep)
        for og, (order, ps_), (vv, ve, wv, cnt) in oracles:
            if learn:
                assert og.learn_call(order, ps_, vv, ve, wv, 1, step, 1.0, 2, 0.01, 1, False, seed, s) == 0
            else:
                assert og.gibbs_dev(order, ps_, vv, wv, cnt, seed, s, True) == 0
        step *= 0.95
        for which, sname, rname in ((_lib.BUF_VALUE, "send", "recv"), (_lib.BUF_VALUE_EVID, "send_evid", "recv_evid")):
            if which == _lib.BUF_VALUE_EVID and not learn:
                continue
            timing["pack_us"].append([_timed(L, p.h, lambda: _lib.check(L.nsk_exchange_pack(p.h, which))) for p in parts])
            torch.cuda.synchronize()
            for q in parts:                                  # the all-gather
                for r, p in enumerate(parts):
                    getattr(q, rname)[r * slot:(r + 1) * slot] = getattr(p, sname)
            torch.cuda.synchronize()
            timing["unpack_us"].append([_timed(L, p.h, lambda: _lib.check(L.nsk_exchange_unpack(p.h, which))) for p in parts])
        for r in range(WORLD):                               # oracle side: owners publish their slices
            lo, hi = shard_range(r, WORLD, nvar)
            for q in range(WORLD):
                if q != r:
                    oracles[q][2][0][lo:hi] = oracles[r][2][0][lo:hi]
                    oracles[q][2][1][lo:hi] = oracles[r][2][1][lo:hi]
        if learn:                                            # w = w_start + sum of deltas
            total = sum(p.w - s0 for p, s0 in zip(parts, starts))
            for p, s0 in zip(parts, starts):
                p.w.copy_(s0 + total)
            ototal = sum(st[2] - s0 for (_, _, st), s0 in zip(oracles, ostarts))
            for (_, _, st), s0 in zip(oracles, ostarts):
                st[2][:] = s0 + ototal
        torch.cuda.synchronize()
    for r in range(WORLD):
        vv, ve, wv, cnt = oracles[r][2]
        lo, hi = shard_range(r, WORLD, nvar)
        got = parts[r].val.cpu().numpy().astype(np.int64)
        assert np.array_equal(got[lo:hi], vv[lo:hi]), ("owned values differ", r)
        for q in range(WORLD):
            if q != r and len(lists[q]):
                need = np.intersect1d(lists[q], parts[r].fg.ghost_needs())
                assert np.array_equal(got[need], vv[need]), ("ghost values differ", r, q)
        if learn:
            assert np.array_equal(parts[r].val_evid.cpu().numpy().astype(np.int64)[lo:hi], ve[lo:hi])
            assert np.allclose(parts[r].w.cpu().numpy(), wv, rtol=0, atol=1e-15)
        else:
            parts[r].fg._pull(0, 0)
            cs = parts[r].fg.cstart
            assert np.array_equal(parts[r].fg.count[cs[lo]:cs[hi]], cnt[cs[lo]:cs[hi]])
    out = {"config": "10M grid (2500x4000) in 8 range shards, all on one MI355X, %s" % ("learning" if learn else "inference"),
           "boundary_values_per_rank": slot,
           "per_shard_us": {k: {"mean": float(np.mean(v[1:] if len(v) > 1 else v)), "max": float(np.max(v[1:] if len(v) > 1 else v))}
                            for k, v in timing.items() if v},
           "note": "one sweep of one shard's kernels alone on the device (HIP events on the library stream); "
                   "first sweep excluded"}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/config4_shards_%s.json" % ("learn" if learn else "inference"), "w") as f:
        json.dump(out, f, indent=1)
