"""N > 1 path on CPU: world_size-2 `gloo` processes run the SAME host code the GPU ranks run
(shard formula, host-only graph planning with an owned range, ghost-need gathering and boundary
planning, the all-gather of the boundary buffers, merge_weight_deltas -- all from
numbskull_amd/distributed.py); the per-variable compute and the pack/unpack kernels are stood in
by the CPU oracle and numpy (tests may use the oracle).  The result must equal a single-process emulation of the
partitioned semantics: every rank samples its own range against the values the others had at the
end of the previous sweep."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import graphgen, session, oracle_of, phases_from_colors, check_coloring, free_port
from numbskull_amd.distributed import (shard_range, plan_boundaries, gather_needs,
                                       merge_weight_deltas)


def _free_port():
    return free_port()          # (outside the ephemeral range: util.free_port)


def _graph(kind):
    if kind == "grid":
        return graphgen.ising_grid(16, 16, weight=0.4)
    if kind == "ragged":       # 225 variables: shards of 112 and 113 -> broadcast fallback
        return graphgen.ising_grid(15, 15, weight=0.4)
    rng = np.random.default_rng(2)
    return graphgen.ising_grid(12, 12, weight=0.1, fixed=False, two_weights=True,
                               evidence=rng.integers(0, 2, 144))


def _rank_sweeps(rank, world, kind, nsweeps, learn, exchange):
    """What one rank does; `exchange(values, values_evid, weights, start)` is the collective step."""
    g = _graph(kind)
    nvar = len(g[1])
    ns, fg = session(g, seed=21)
    fg.own_range = shard_range(rank, world, nvar)
    color, info = fg.plan()
    lo, hi = fg.own_range
    assert np.all(color[:lo] == -1) and np.all(color[hi:] == -1) and np.all(color[lo:hi] >= 0)
    assert info["nowned"] == hi - lo
    check_coloring(fg, color)
    order, ps = phases_from_colors(color)
    og = oracle_of(fg)
    vv, ve, wv, cnt = og.initial_state()
    step = 0.01
    for s in range(nsweeps):
        start = wv.copy()
        if learn:
            assert og.learn_dev(order, ps, vv, ve, wv, step, 2, 0.01, 1, False, 21, s) == 0
            step *= 0.9
        else:
            assert og.gibbs_dev(order, ps, vv, wv, cnt, 21, s, True) == 0
        exchange(vv, ve, wv, start)
    return vv, ve, wv, cnt


def _worker(rank, world, port, kind, nsweeps, learn, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    g = _graph(kind)
    nvar = len(g[1])
    ns, fg = session(g, seed=21)
    fg.own_range = shard_range(rank, world, nvar)
    needs = fg.ghost_needs(host_only=True)
    lo, hi = fg.own_range
    assert np.all((needs < lo) | (needs >= hi))
    lists, slot = plan_boundaries(gather_needs(dist, torch, needs, world, "cpu"), world, nvar)
    assert all(np.all((b >= shard_range(r, world, nvar)[0]) & (b < shard_range(r, world, nvar)[1]))
               for r, b in enumerate(lists))
    assert set(needs.tolist()) <= set(np.concatenate(lists).tolist())

    def exchange(vv, ve, wv, start):
        for arr in (vv, ve) if learn else (vv,):
            send = torch.zeros(slot, dtype=torch.int8)                      # nsk_exchange_pack
            send[:len(lists[rank])] = torch.from_numpy(arr[lists[rank]].astype(np.int8))
            recv = torch.zeros(slot * world, dtype=torch.int8)
            dist.all_gather_into_tensor(recv, send)
            for src in range(world):                                        # nsk_exchange_unpack
                if src != rank:
                    arr[lists[src]] = recv[src * slot:src * slot + len(lists[src])].numpy()
        if learn:
            tw = torch.from_numpy(wv)
            merge_weight_deltas(dist, tw, torch.from_numpy(start))
    vv, ve, wv, cnt = _rank_sweeps(rank, world, kind, nsweeps, learn, exchange)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), vv=vv, ve=ve, wv=wv, cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


def _emulate(world, kind, nsweeps, learn):
    """Single-process statement of the same semantics (stale ghosts, delta-sum weights)."""
    g = _graph(kind)
    nvar = len(g[1])
    ranks = []
    for r in range(world):
        ns, fg = session(g, seed=21)
        fg.own_range = shard_range(r, world, nvar)
        color, _ = fg.plan()
        og = oracle_of(fg)
        ranks.append((og, phases_from_colors(color), og.initial_state(), fg.own_range))
    step = 0.01
    for s in range(nsweeps):
        starts = [st[2].copy() for _, _, st, _ in ranks]
        for og, (order, ps), (vv, ve, wv, cnt), _ in ranks:
            if learn:
                og.learn_dev(order, ps, vv, ve, wv, step, 2, 0.01, 1, False, 21, s)
            else:
                og.gibbs_dev(order, ps, vv, wv, cnt, 21, s, True)
        step *= 0.9
        for _, _, (vv, ve, wv, cnt), (lo, hi) in ranks:          # owners publish their slices
            for _, _, (vv2, ve2, _, _), _ in ranks:
                vv2[lo:hi] = vv[lo:hi]
                ve2[lo:hi] = ve[lo:hi]
        if learn:
            total = sum(st[2] - s0 for (_, _, st, _), s0 in zip(ranks, starts))
            for (_, _, st, _), s0 in zip(ranks, starts):
                st[2][:] = s0 + total
    return ranks


@pytest.mark.parametrize("kind,learn", [("grid", False), ("ragged", False), ("learn", True)])
def test_two_rank_gloo_matches_emulation(tmp_path, kind, learn):
    world, nsweeps = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), kind, nsweeps, learn, str(tmp_path)), nprocs=world,
             join=True)
    ranks = _emulate(world, kind, nsweeps, learn)
    got = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        _, _, (vv, ve, wv, cnt), (lo, hi) = ranks[r]
        # only boundary values travel, so a replica's copy of another rank's INTERIOR variables is
        # stale by design; every owner's slice must match the emulation
        assert np.array_equal(got[r]["vv"][lo:hi], vv[lo:hi]), (kind, r)
        assert np.array_equal(got[r]["cnt"], cnt)
        if learn:
            assert np.array_equal(got[r]["ve"][lo:hi], ve[lo:hi])
            assert np.allclose(got[r]["wv"], wv, rtol=0, atol=1e-15)
    if learn:
        assert np.array_equal(got[0]["wv"], got[1]["wv"])
        assert np.any(got[0]["wv"] != 0.1)


def test_shard_formula_is_the_references():
    # inference.py:17-18: start = shardID*nvar//nshards, end = (shardID+1)*nvar//nshards
    for nvar in (0, 1, 7, 225, 10 ** 7):
        for world in (1, 2, 3, 8):
            b = [shard_range(r, world, nvar) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nvar
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all((r * nvar) // world == b[r][0] for r in range(world))


def test_plan_colours_every_kind_of_graph():
    for g, hbv in ((graphgen.ising_grid(9, 14), False), (graphgen.ising_pairs(30), False),
                   (graphgen.lf_graph(0.1, [1.0, 0.5, 0.2], 6), False),
                   (graphgen.mixed_lr_graph(2000, seed=8), True)):
        ns, fg = session(g, head_by_vid=hbv)
        color, info = fg.plan()
        assert color.min() >= 0 and info["nowned"] == len(g[1])
        assert info["ncolors"] == color.max() + 1
        check_coloring(fg, color, hbv)
    g = graphgen.ising_grid(50, 40)
    color, info = session(g)[1].plan()
    degree = 2.0 * len(g[2]) / 2000
    # B_inf = 2 + 4 + degree*23 + 1 + 8 (106.9 B/update at degree 3.996)
    assert info["ncolors"] == 2
    assert abs(info["alg_bytes_inference"] / 2000 - (15 + 23 * degree)) < 1e-9


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torch.distributed.run starts the ranks itself as a child
    process, rendezvous over gloo, plans both partitions and prints ONE JSON line (dry run: the part
    of the N-rank launch that needs no GPU)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSK_BENCH_DRYRUN="1", NSK_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--workload", "ising64k",
                        "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["config"]["sampled_total"] == 256 * 256
    assert out["config"]["boundary_total"] == 2 * 256           # one grid row on each side of the cut


def test_bench_dry_run_generates_lr_shards_alone():
    """The LR workloads of an N-rank run never materialise the whole graph: every rank generates its
    shard (graphgen.mixed_lr_shard) and the dry run reports the largest peak RSS over the ranks."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSK_BENCH_DRYRUN="1", NSK_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--workload", "lr300k_learn",
                        "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["config"]["sampled_total"] == 300000
    assert 150000 < out["config"]["variables_held_by_rank0"] < 170000      # its half + the ghosts it reads
    assert 0 < out["config"]["peak_rss_gb_max_over_ranks"] < 8


def test_lr_shard_generator_equals_the_cut_of_the_whole_graph():
    """graphgen.mixed_lr_shard(nvar, lo, hi) == graphgen.extract_shard(graphgen.mixed_lr_graph(nvar), lo, hi),
    array for array, for shards at both ends, in the middle, across generator blocks, and an empty one."""
    from numbskull_amd import graphgen
    nvar = 40000
    g = graphgen.mixed_lr_graph(nvar, seed=5, block=4096)
    assert (g[3]["vid"] == 0).sum() > 100 and (g[3]["vid"] == nvar - 1).sum() > 100      # the clipped windows' hubs
    for lo, hi in ((0, 5000), (5000, 10000), (35000, 40000), (4090, 4100), (77, 77)):
        a, gids, own = graphgen.mixed_lr_shard(nvar, lo, hi, seed=5, block=4096)
        b, gids2, own2 = graphgen.extract_shard(g, lo, hi)
        assert np.array_equal(gids, gids2) and own == own2 and a[5] == b[5]
        for x, y in zip(a[:5], b[:5]):
            assert x.dtype == y.dtype and np.array_equal(x, y)
    # several shards in one pass over the blocks (a process that holds them all: the 8-handle tests)
    ranges = ((0, 5000), (5000, 10000), (35000, 40000), (4090, 4100), (77, 77))
    for (lo, hi), (a, gids, own) in zip(ranges, graphgen.mixed_lr_shards(nvar, ranges, seed=5, block=4096)):
        b, gids2, own2 = graphgen.mixed_lr_shard(nvar, lo, hi, seed=5, block=4096)
        assert np.array_equal(gids, gids2) and own == own2 and a[5] == b[5]
        for x, y in zip(a[:5], b[:5]):
            assert x.dtype == y.dtype and np.array_equal(x, y)
    # a different block size is a different graph (the streams are keyed per block), the same one is not
    assert not np.array_equal(graphgen.mixed_lr_graph(nvar, seed=5, block=8192)[3], g[3])
    assert np.array_equal(graphgen.mixed_lr_graph(nvar, seed=5, block=4096)[3], g[3])


def test_pairwise_boundary_lists():
    """distributed.plan_pairs: rank s sends rank d exactly what d reads from s; the peer-to-peer set-up
    arguments of every rank are mutually consistent (segment bases and totals)."""
    from numbskull_amd.distributed import plan_pairs, shard_range
    world, nvar = 4, 1000
    rng = np.random.default_rng(1)
    needs = []
    for r in range(world):
        lo, hi = shard_range(r, world, nvar)
        cand = np.setdiff1d(np.arange(nvar), np.arange(lo, hi))
        needs.append(np.sort(rng.choice(cand, 60, replace=False)).astype(np.int32))
    needs[2] = needs[2][(needs[2] < 250)]                # rank 2 reads from rank 0 only
    pairs = plan_pairs(needs, world, nvar)
    for d in range(world):
        assert len(pairs[d][d]) == 0
        assert np.array_equal(np.concatenate([pairs[s][d] for s in range(world)]), needs[d])
        for s_ in range(world):
            lo, hi = shard_range(s_, world, nvar)
            assert ((pairs[s_][d] >= lo) & (pairs[s_][d] < hi)).all()
    assert all(len(pairs[s_][2]) == 0 for s_ in (1, 3))


def test_shard_local_graph_plans_like_the_whole_graph():
    """graphgen.extract_shard: a rank that holds only its shard (owned variables, the ghosts they read
    flagged isEvidence 4, the factors that touch them, renumbered in ascending global order -- what
    the reference's minions load, salt/src/numbskull_minion.py:185) colours its variables and lists
    its ghost needs exactly like a handle that is given the whole graph with own_range."""
    import numbskull_amd
    from numbskull_amd import graphgen
    from numbskull_amd.distributed import shard_range
    for g, hbv in ((graphgen.ising_grid(40, 30, weight=0.3), False),
                   (graphgen.mixed_lr_graph(4000, seed=12, nweights=300), True)):
        nvar = len(g[1])
        total = 0
        for r in range(3):
            lo, hi = shard_range(r, 3, nvar)
            ns = numbskull_amd.NumbSkull(quiet=True, seed=1, head_by_vid=hbv)
            ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                               own_range=(lo, hi))
            whole = ns.factorGraphs[0]
            sg, gids, (l0, l1) = graphgen.extract_shard(g, lo, hi)
            assert len(gids) < nvar and np.array_equal(gids[l0:l1], np.arange(lo, hi))
            assert (sg[1]["isEvidence"][:l0] == 4).all() and (sg[1]["isEvidence"][l1:] == 4).all()
            ns2 = numbskull_amd.NumbSkull(quiet=True, seed=1, head_by_vid=hbv)
            ns2.loadFactorGraph(*sg[:5], int(sg[5]), own_range=(l0, l1), global_ids=gids)
            part = ns2.factorGraphs[0]
            cw, cl = whole.plan()[0], part.plan()[0]
            assert np.array_equal(cw[lo:hi], cl[l0:l1]) and (cl[:l0] == -1).all() and (cl[l1:] == -1).all()
            assert np.array_equal(whole.ghost_needs(host_only=True), gids[part.ghost_needs(host_only=True)])
            total += len(sg[2])
        assert total < 2 * len(g[2])                      # factors are shared by at most the shards they touch


def _merge_worker(rank, world, port, nw, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from numbskull_amd.distributed import merge_weight_deltas_sliced
    rng = np.random.default_rng(100)
    start = rng.normal(size=nw)
    mine = start + np.random.default_rng(200 + rank).normal(size=nw) * 10.0 ** np.random.default_rng(300 + rank).integers(-12, 3, nw)
    w = torch.from_numpy(mine.copy())
    merge_weight_deltas_sliced(dist, w, torch.from_numpy(start), rank, world)
    np.save(os.path.join(outdir, "w%d.npy" % rank), w.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nw", [(3, 7), (3, 2), (4, 10), (4, 1000003 % 4099)])
def test_sliced_weight_merge_on_uneven_slices(tmp_path, world, nw):
    """The reduce-scatter + all-gather merge of the weight deltas (the peer-to-peer path's, nsk_kernels_misc.h
    p2p_reduce_slice; salt/src/numbskull_master.py:223-224 w = w0 + sum of the minions' deltas) with world sizes that are
    not powers of two and weight counts they do not divide -- slices of unequal length, an empty slice (2 weights on 3
    ranks): every rank ends with the SAME bits, equal to the rank-ordered sum."""
    from numbskull_amd.distributed import weight_slice
    mp.spawn(_merge_worker, args=(world, _free_port(), nw, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(100)
    start = rng.normal(size=nw)
    deltas = []
    for r in range(world):
        mine = start + np.random.default_rng(200 + r).normal(size=nw) * 10.0 ** np.random.default_rng(300 + r).integers(-12, 3, nw)
        deltas.append(mine - start)
    want = np.zeros(nw)
    for q in range(world):
        lo, hi = weight_slice(q, world, nw), weight_slice(q + 1, world, nw)
        t = deltas[0][lo:hi].copy()
        for r in range(1, world):
            t += deltas[r][lo:hi]
        want[lo:hi] = start[lo:hi] + t
    got = [np.load(tmp_path / ("w%d.npy" % r)) for r in range(world)]
    for r in range(world):
        assert np.array_equal(got[r], want), r
    assert [weight_slice(q, world, nw) for q in range(world + 1)][-1] == nw


@pytest.mark.parametrize("two,ev", [(False, False), (True, True)])
def test_grid_shard_generator_equals_the_cut_of_the_whole_grid(two, ev):
    """graphgen.ising_grid_shard builds a rank's shard of a grid from its own cells (+ one row): record for record what
    extract_shard cuts out of the whole grid -- ragged ranges that split rows, the first and the last shard, an empty one."""
    n, m = 23, 17
    rng = np.random.default_rng(4)
    evid = rng.integers(0, 2, n * m)
    whole = graphgen.ising_grid(n, m, weight=0.3, fixed=not two, two_weights=two, evidence=evid if ev else None)
    for lo, hi in [(0, 50), (50, 51), (51, 200), (200, 391), (100, 100), (0, 391), (380, 391), (3, 20)]:
        a, ga, oa = graphgen.extract_shard(whole, lo, hi)
        b, gb, ob = graphgen.ising_grid_shard(n, m, lo, hi, weight=0.3, fixed=not two, two_weights=two,
                                              evidence=(lambda ids: evid[ids]) if ev else None)
        assert np.array_equal(ga, gb) and oa == ob, (lo, hi)
        for x, y in zip(a[:5], b[:5]):
            assert x.dtype == y.dtype and np.array_equal(x, y), (lo, hi)
        assert a[5] == b[5]
