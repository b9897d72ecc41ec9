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


def test_median_refined_order_recovers_a_shuffled_lr_graph():
    """Config #5's generator: 99 % of a factor's members within +-1024 ids, 1 % anywhere.  A breadth-first walk follows
    the long edges; the maximum-adjacency walk refined by median-of-neighbours placement finds the band again."""
    g = graphgen.mixed_lr_graph(60000, seed=7)
    n, parts = len(g[1]), 8
    native = partition.comm_volume(n, g[2], g[3], parts)
    gs, _ = shuffled(g, 2)
    lost = partition.comm_volume(n, gs[2], gs[3], parts)
    part, order = partition.find_parts(n, gs[2], gs[3], parts)            # "auto": the best of ids / bfs / median
    found = partition.comm_volume(n, gs[2], gs[3], parts, order)
    bfs = partition.comm_volume(n, gs[2], gs[3], parts, partition.graph_order(n, gs[2], gs[3], "bfs")[0])
    assert lost > 10 * native and found < lost / 8 and found < 2.2 * native and found < bfs / 3
    assert np.array_equal(np.bincount(part, minlength=parts), [shard_range(r, parts, n)[1] - shard_range(r, parts, n)[0] for r in range(parts)])
    # "never worse than what came in": on the generator's own ids the answer is those ids (or better)
    _, keep = partition.find_parts(n, g[2], g[3], parts)
    assert partition.comm_volume(n, g[2], g[3], parts, keep) <= native


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
