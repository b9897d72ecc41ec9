"""Graph-aware partitioning in front of the range partition (the reference's
find_connected_components / find_metis_parts, salt/src/messages.py:542-670): host logic, no GPU."""
import numpy as np

from numbskull_amd import graphgen, partition
from numbskull_amd.distributed import shard_range


def shuffled(g, seed):
    """The same graph with its variable ids permuted at random (what a loader that numbers variables in arrival
    order hands over)."""
    n = len(g[1])
    perm = np.random.default_rng(seed).permutation(n)        # order[new] = old
    return partition.relabel(g, perm), perm


def members(g):
    """Factor -> frozenset of member ids: the graph's structure."""
    f, fm = g[2], g[3]
    return [frozenset(fm["vid"][o:o + a].tolist()) for o, a in zip(f["ftv_offset"], f["arity"])]


def test_connected_components_are_found_and_kept_together():
    a = graphgen.ising_grid(6, 7, weight=0.3)
    b = graphgen.ising_grid(5, 4, weight=0.3)
    # two grids side by side in one graph, then every id shuffled
    w, va, fa, ma, da, ea = a
    _, vb, fb, mb, db, eb = b
    fb2, mb2 = fb.copy(), mb.copy()
    fb2["ftv_offset"] += int(ea)
    mb2["vid"] += len(va)
    g = (w, np.concatenate([va, vb]), np.concatenate([fa, fb2]), np.concatenate([ma, mb2]),
         np.concatenate([da, db]), int(ea) + int(eb))
    gs, perm = shuffled(g, 5)
    n = len(gs[1])
    cc = partition.find_connected_components(n, gs[2], gs[3])
    assert len(np.unique(cc)) == 2
    old = perm                                                 # new id i is old id perm[i]
    assert len(np.unique(cc[old < len(va)])) == 1 and len(np.unique(cc[old >= len(va)])) == 1
    order, cc2, ncc = partition.graph_order(n, gs[2], gs[3], "components")
    assert ncc == 2 and np.array_equal(cc, cc2) and np.array_equal(np.sort(order), np.arange(n))
    first = cc[order[0]]
    k = int((cc == first).sum())
    assert (cc[order[:k]] == first).all() and (cc[order[k:]] != first).all()       # one component after the other
    assert partition.comm_volume(n, gs[2], gs[3], 2, order) <= partition.comm_volume(n, gs[2], gs[3], 2)


def test_breadth_first_order_recovers_the_locality_of_a_shuffled_grid():
    rows, cols, parts = 48, 40, 8
    g = graphgen.ising_grid(rows, cols, weight=0.2)
    n = rows * cols
    native = partition.comm_volume(n, g[2], g[3], parts)
    assert native == 2 * (parts - 1) * cols                   # one grid row on each side of every cut
    gs, _ = shuffled(g, 11)
    lost = partition.comm_volume(n, gs[2], gs[3], parts)
    assert lost > 0.8 * n                                     # a shuffled id order cuts nearly every variable off a neighbour
    part, order = partition.find_parts(n, gs[2], gs[3], parts)
    found = partition.comm_volume(n, gs[2], gs[3], parts, order)
    assert found < 4 * native and found < lost / 10           # BFS fronts of a grid are diagonals: ~ sqrt(2) x a row
    # the parts are exactly the shard formula's
    for r in range(parts):
        lo, hi = shard_range(r, parts, n)
        assert int((part == r).sum()) == hi - lo
        assert (part[order[lo:hi]] == r).all()


def test_relabelling_keeps_the_graph():
    g = graphgen.mixed_lr_graph(3000, seed=4, nweights=40)
    n = len(g[1])
    order, _, _ = partition.graph_order(n, g[2], g[3], "bfs")
    h = partition.relabel(g, order)
    assert len(h[2]) == len(g[2]) and int(h[5]) == int(g[5])
    back = [frozenset(int(order[v]) for v in s) for s in members(h)]
    assert back == members(g)                                 # same factors over the same (renamed) variables
    for name in ("isEvidence", "initialValue", "dataType", "cardinality"):
        assert np.array_equal(h[1][name], g[1][name][order])
    assert np.array_equal(h[3]["dense_equal_to"], g[3]["dense_equal_to"])
    # the generator's ids are already local: the walk may not do much better, and must not do much worse
    assert partition.comm_volume(n, g[2], g[3], 4, order) < 1.5 * partition.comm_volume(n, g[2], g[3], 4)


def test_multilevel_partitioner_recovers_a_shuffled_lr_graph():
    """Config #5's generator: 99 % of a factor's members within +-1024 ids, 1 % anywhere.  A breadth-first walk follows
    the long edges; the median-refined maximum-adjacency walk finds the band again (1.6x the generator's own ids at
    this size); the multilevel partitioner (the reference's find_metis_parts, objtype = vol) gets within 10 %."""
    g = graphgen.mixed_lr_graph(60000, seed=7)
    n, parts = len(g[1]), 8
    native = partition.comm_volume(n, g[2], g[3], parts)
    gs, _ = shuffled(g, 2)
    lost = partition.comm_volume(n, gs[2], gs[3], parts)
    part, order = partition.find_parts(n, gs[2], gs[3], parts)            # "auto": the best of ids / bfs / multilevel
    found = partition.comm_volume(n, gs[2], gs[3], parts, order)
    bfs = partition.comm_volume(n, gs[2], gs[3], parts, partition.graph_order(n, gs[2], gs[3], "bfs")[0])
    median = partition.comm_volume(n, gs[2], gs[3], parts, partition.graph_order(n, gs[2], gs[3], "median")[0])
    assert lost > 10 * native and found <= 1.1 * native and found < median and found < bfs / 3, (native, found, median, bfs)
    assert np.array_equal(np.bincount(part, minlength=parts), [shard_range(r, parts, n)[1] - shard_range(r, parts, n)[0] for r in range(parts)])
    # "never worse than what came in": on the generator's own ids the answer is those ids (or better)
    _, keep = partition.find_parts(n, g[2], g[3], parts)
    assert partition.comm_volume(n, g[2], g[3], parts, keep) <= native


def test_multilevel_partitioner_sizes_determinism_and_odd_inputs():
    """nsk_graph_partition: the parts are exactly the shard formula's (any part count, also more parts than
    variables), the answer depends on (graph, parts, seed) only, the reported volume is nsk_comm_volume's, isolated
    variables and separate components are handled, and member ids outside the graph are refused."""
    import pytest
    g = graphgen.mixed_lr_graph(20000, seed=3)
    n = len(g[1])
    gs, _ = shuffled(g, 9)
    for parts in (1, 2, 3, 7, 8, 13):
        order, st = partition.multilevel_order(n, gs[2], gs[3], parts)
        assert np.array_equal(np.sort(order), np.arange(n))
        if parts > 1:
            assert st["volume"] == partition.comm_volume(n, gs[2], gs[3], parts, order) <= st["volume_before"]
        order2, _ = partition.multilevel_order(n, gs[2], gs[3], parts)
        assert np.array_equal(order, order2)
    a, _ = partition.multilevel_order(n, gs[2], gs[3], 8, seed=1)
    b, _ = partition.multilevel_order(n, gs[2], gs[3], 8, seed=2)
    assert not np.array_equal(a, b)                            # (another seed, another matching)
    # two grids and a handful of variables no factor touches; 5 parts
    ga = graphgen.ising_grid(30, 20, weight=0.3)
    gb = graphgen.ising_grid(10, 25, weight=0.3)
    w, va, fa, ma, da, ea = ga
    _, vb, fb, mb, db, eb = gb
    fb2, mb2 = fb.copy(), mb.copy()
    fb2["ftv_offset"] += int(ea)
    mb2["vid"] += len(va)
    lone = va[:37].copy()
    both = (w, np.concatenate([va, vb, lone]), np.concatenate([fa, fb2]), np.concatenate([ma, mb2]),
            np.concatenate([da, db, da[:37]]), int(ea) + int(eb))
    gs2, _ = shuffled(both, 4)
    n2 = len(gs2[1])
    part, order = partition.find_parts(n2, gs2[2], gs2[3], 5, method="multilevel")
    assert np.array_equal(np.bincount(part, minlength=5), [shard_range(r, 5, n2)[1] - shard_range(r, 5, n2)[0] for r in range(5)])
    assert partition.comm_volume(n2, gs2[2], gs2[3], 5, order) < 400      # (887 variables: a few grid rows' worth)
    # more parts than variables
    tiny = graphgen.ising_grid(2, 3, weight=0.1)
    part, order = partition.find_parts(6, tiny[2], tiny[3], 8, method="multilevel")
    assert np.array_equal(np.bincount(part, minlength=8), [shard_range(r, 8, 6)[1] - shard_range(r, 8, 6)[0] for r in range(8)])
    bad = tiny[3].copy()
    bad["vid"][0] = 6
    with pytest.raises(IndexError):
        partition.multilevel_order(6, tiny[2], bad, 2)


_ORDER_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from numbskull_amd import graphgen, partition
g = graphgen.mixed_lr_graph(40000, seed=5)
g = partition.relabel(g, np.random.default_rng(1).permutation(len(g[1])))
order, st = partition.multilevel_order(len(g[1]), g[2], g[3], 8)
print(hashlib.sha256(order.tobytes()).hexdigest(), st["volume"])
"""


def test_multilevel_partitioner_is_independent_of_the_thread_count():
    """The graph build, the median rounds and the gain evaluation of the volume refinement run over host threads
    (static index blocks against a frozen state): the order must not depend on how many there are."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for threads in ("1", "5"):
        r = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT, repo], env=dict(os.environ, NSK_COMPILE_THREADS=threads),
                           capture_output=True, text=True, timeout=600, cwd=repo)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def test_multilevel_partitioner_on_a_shuffled_grid():
    """A mesh: the recursive bisection of the coarsest graph cuts blocks, not strips -- fewer values across the cuts than
    the row-major ids' 7 straight cuts."""
    rows, cols, parts = 120, 100, 8
    g = graphgen.ising_grid(rows, cols, weight=0.2)
    n = rows * cols
    native = partition.comm_volume(n, g[2], g[3], parts)
    gs, _ = shuffled(g, 11)
    order, st = partition.multilevel_order(n, gs[2], gs[3], parts)
    assert partition.comm_volume(n, gs[2], gs[3], parts, order) < native


def test_partial_factor_rewriting_keeps_every_factor_value():
    """f3 host logic (graphgen.partial_factors; the reference's PF surgery, salt/src/messages.py:1083-1206): in every
    shard, a clause whose foreign members were replaced by aggregates has -- for any assignment -- the value it has
    with the members themselves, and the shard reads fewer foreign values."""
    g = graphgen.voter_graph(300, width=9, seed=1)
    n, world = len(g[1]), 4
    rng = np.random.default_rng(0)
    x = rng.integers(0, 2, n)                                  # one global assignment

    def clause_value(func, vals):                              # inference.py:177-200
        if func == 1:
            return 1 if (vals == 1).any() else -1
        return -1 if (vals == 0).any() else 1
    total_before = total_after = 0
    for r in range(world):
        lo, hi = shard_range(r, world, n)
        sh, gids, own = graphgen.extract_shard(g, lo, hi)
        sh2, gids2, own2, pf = graphgen.partial_factors(sh, gids, own, n, world)
        assert own2 == own and np.array_equal(gids2[:len(gids)], gids) and (gids2[len(gids):] >= n).all()
        xl = np.zeros(len(gids2), np.int64)
        xl[:len(gids)] = x[gids]
        for q, op, members, lid in pf:
            assert all(shard_range(q, world, n)[0] <= m < shard_range(q, world, n)[1] for m in members) and q != r
            xl[lid] = int((x[members] == 1).any()) if op == 0 else int((x[members] != 0).all())
            assert sh2[1]["isEvidence"][lid] == 4 and sh2[1]["cardinality"][lid] == 2
        f1, m1, f2, m2 = sh[2], sh[3], sh2[2], sh2[3]
        assert len(f1) == len(f2) and int(f2["arity"].sum()) == len(m2) == int(sh2[5])
        for i in range(len(f1)):
            a = x[gids[m1["vid"][f1["ftv_offset"][i]:f1["ftv_offset"][i] + f1["arity"][i]]]]
            b = xl[m2["vid"][f2["ftv_offset"][i]:f2["ftv_offset"][i] + f2["arity"][i]]]
            fn = int(f1["factorFunction"][i])
            assert clause_value(fn, a) == clause_value(fn, b), (r, i)
        used1 = np.unique(m1["vid"]); used2 = np.unique(m2["vid"])
        total_before += int(((used1 < own[0]) | (used1 >= own[1])).sum())
        total_after += int(((used2 < own[0]) | (used2 >= own[1])).sum())
    assert total_after < total_before / 3
