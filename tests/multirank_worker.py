"""Worker of tests/test_multirank_gpu.py: one of WORLD_SIZE processes (all on cuda:0, gloo
rendezvous) running the real multi-rank host path -- FactorGraph with an owned range, ghost-need
gathering, boundary planning, PartitionedSampler's per-sweep loop -- against the oracle's emulation
of the partitioned semantics.  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import graphgen, oracle_of, phases_from_colors          # noqa: E402
import numbskull_amd                                              # noqa: E402
from numbskull_amd.distributed import PartitionedSampler, shard_range   # noqa: E402


def graph(kind):
    if kind == "grid":
        rng = np.random.default_rng(3)
        return graphgen.ising_grid(40, 30, weight=0.1, fixed=False, two_weights=True,
                                   evidence=rng.integers(0, 2, 1200)), False
    return graphgen.mixed_lr_graph(4000, seed=12, nweights=300), True     # general tiles, global atomics


def main():
    kind, learn, nsweeps = sys.argv[1], sys.argv[2] == "learn", 4
    if len(sys.argv) > 3 and sys.argv[3].startswith("p2p") and not learn:
        nsweeps = 87                              # long enough for captured sweep sequences of both sizes (grids: 1 + 64 + 16 + 6)
    local = len(sys.argv) > 3 and sys.argv[3] in ("local", "p2plocal")   # every rank holds only its shard (+ ghosts)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, hbv = graph(kind)
    nvar = len(g[1])
    handles, gids = [], []
    for r in range(world):                       # rank r's graph; the others only to emulate them
        ns = numbskull_amd.NumbSkull(quiet=True, seed=31, head_by_vid=hbv)
        if local:
            sg, ids, own = graphgen.extract_shard(g, *shard_range(r, world, nvar))
            ns.loadFactorGraph(*sg[:5], int(sg[5]), own_range=own, global_ids=ids)
            assert len(ids) < nvar                                   # the shard is a part of the graph
        else:
            w, v, f, fm, dm, edges = [x.copy() if isinstance(x, np.ndarray) else x for x in g]
            ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=shard_range(r, world, nvar))
            ids = np.arange(nvar, dtype=np.int64)
        handles.append(ns.factorGraphs[0])
        gids.append(ids)
    fg = handles[rank]
    p2p = len(sys.argv) > 3 and sys.argv[3].startswith("p2p")   # boundary values (learning: both chains + weight deltas) written into the peer's memory (hipIpc)
    if p2p:                                               # a stream of its own: sweep sequences can be captured
        torch.cuda.set_stream(torch.cuda.Stream())        # (the legacy default stream cannot)
    sampler = PartitionedSampler(fg, dist, torch, rank, world, nvar_global=nvar, p2p=p2p)   # native RCCL refuses one device
    assert not sampler.native                                        # for two ranks: torch loop
    assert sampler.p2p == p2p, "peer-to-peer exchange could not be set up"
    oracles = []
    for r in range(world):
        color = handles[r].plan()[0] if r != rank else fg.colors()
        og = oracle_of(handles[r], hbv)
        oracles.append((og, phases_from_colors(color), og.initial_state()))
    if learn:
        sampler.learn(nsweeps, 0.01, 0.9, 2, 0.01, 1)
    else:
        sampler.gibbs(nsweeps, True, False)
    sampler.check()                              # a peer-to-peer exchange that timed out raises here
    torch.cuda.synchronize()

    def loc(r, ids):                             # global ids -> rank r's ids (all present)
        at = np.searchsorted(gids[r], ids)
        assert np.array_equal(gids[r][at], ids)
        return at

    step = 0.01
    for s in range(nsweeps):
        starts = [st[2].copy() for _, _, st in oracles]
        for og, (order, ps), (vv, ve, wv, cnt) in oracles:
            if learn:
                assert og.learn_call(order, ps, vv, ve, wv, 1, step, 1.0, 2, 0.01, 1, False, 31, s) == 0
            else:
                assert og.gibbs_dev(order, ps, vv, wv, cnt, 31, s, True) == 0
        step *= 0.9
        for r in range(world):                   # owners publish their boundary values
            for q in range(world):
                if q != r:
                    b = np.intersect1d(sampler.lists[r], gids[q])    # what q holds of r's boundary
                    oracles[q][2][0][loc(q, b)] = oracles[r][2][0][loc(r, b)]
                    oracles[q][2][1][loc(q, b)] = oracles[r][2][1][loc(r, b)]
        if learn:
            total = sum(st[2] - s0 for (_, _, st), s0 in zip(oracles, starts))
            for (_, _, st), s0 in zip(oracles, starts):
                st[2][:] = s0 + total
    vv, ve, wv, cnt = oracles[rank][2]
    lo, hi = fg.own_range
    got = sampler.val.cpu().numpy().astype(np.int64)
    assert np.array_equal(got[lo:hi], vv[lo:hi]), "owned values differ"
    needs = sampler.global_needs()
    for r in range(world):
        if r != rank:
            b = np.intersect1d(sampler.lists[r], needs)
            assert np.array_equal(got[loc(rank, b)], vv[loc(rank, b)]), "ghost values differ"
    if learn:
        gote = sampler.val_evid.cpu().numpy().astype(np.int64)
        assert np.array_equal(gote[lo:hi], ve[lo:hi]), "evidence-chain values differ"
        if p2p:       # w_start + (d_0 + d_1 + ...) in rank order on one owner per weight: the emulation's very sums
            assert np.array_equal(sampler.w.cpu().numpy(), wv), "weights differ"
        else:         # (a collective all-reduce does not promise an order of additions)
            assert np.allclose(sampler.w.cpu().numpy(), wv, rtol=0, atol=1e-13), "weights differ"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
