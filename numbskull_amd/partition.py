"""Graph-aware partitioning in front of the range partition.

The reference's distributed prototype partitions a factor graph for its minions by connected components
(``find_connected_components``, salt/src/messages.py:542-590) or with METIS, objective communication volume
(``find_metis_parts``, messages.py:593-670), and stores ``variable -> part``.  The samplers here shard by the
reference's own formula over variable ids, ``[g*n//G, (g+1)*n//G)`` (inference.py:17-18), so a graph-aware
partition is a VARIABLE ORDER in front of it: connected components stay together and, inside a component,
variables follow a breadth-first (Cuthill-McKee) walk from a pseudo-peripheral variable -- the cut of the shard
formula then runs along a few BFS fronts instead of through whatever order the ids came in, and the parts have
exactly the formula's sizes (a METIS part is balanced within a tolerance).  ``relabel`` rewrites a graph in the
new ids; every downstream piece -- ``FactorGraph(own_range=...)``, ``graphgen.extract_shard``, the peer-to-peer
exchange -- works on the relabelled graph unchanged, and ``order`` maps results back.

``method="multilevel"`` is the partitioner proper (``nsk_graph_partition``, csrc/nsk_partition.cpp): heavy-edge
matching, a partition of the coarse graph, k-way boundary refinement on the way up and a last refinement of the
communication volume itself with the parts held at exactly the shard formula's sizes -- what the reference asks
METIS for (``objtype = vol``).

Native: ``nsk_graph_order`` / ``nsk_comm_volume`` (csrc/nsk_host.cpp), ``nsk_graph_partition``
(csrc/nsk_partition.cpp); O(edges) per level, host threads, no GPU.
"""

import ctypes as C

import numpy as np

from . import _lib
from .numbskulltypes import Factor, FactorToVar


def _records(factor, fmap):
    return _lib.as_c(factor, Factor), _lib.as_c(fmap, FactorToVar)


def graph_order(nvar, factor, fmap, method="bfs"):
    """``(order, cc_id, ncc)``: ``order[new id] = old id``; ``cc_id[old id]`` = connected component (numbered by
    smallest member id).  ``method``: "components" (components together, ids ascending inside), "bfs"
    (breadth-first order inside: meshes), "mas" (maximum-adjacency order inside) or "median" (the maximum-adjacency
    walk refined by rounds of median-of-neighbours placement: local structure laced with a few long edges, such
    as config #5's)."""
    f, m = _records(factor, fmap)
    order = np.empty(int(nvar), np.int64)
    cc = np.empty(int(nvar), np.int64)
    ncc = C.c_int64()
    _lib.check(_lib.lib().nsk_graph_order(int(nvar), len(f), _lib.ptr(f), len(m), _lib.ptr(m),
                                          {"components": 0, "bfs": 1, "mas": 2, "median": 3}[method], _lib.ptr(order), _lib.ptr(cc), C.byref(ncc)))
    return order, cc, int(ncc.value)


def multilevel_order(nvar, factor, fmap, parts, seed=1):
    """``(order, stats)`` of the multilevel partitioner for ``parts`` shards: ``order[new id] = old id`` and range g of the
    new ids (the shard formula's) is part g.  ``stats``: levels, coarsest vertices, edge cut after uncoarsening,
    communication volume before / after the last refinement.  Deterministic in (graph, parts, seed)."""
    f, m = _records(factor, fmap)
    order = np.empty(int(nvar), np.int64)
    stats = np.zeros(5, np.int64)
    _lib.check(_lib.lib().nsk_graph_partition(int(nvar), len(f), _lib.ptr(f), len(m), _lib.ptr(m), int(parts), int(seed),
                                              _lib.ptr(order), C.c_void_p(0), _lib.ptr(stats)))
    return order, dict(zip(("levels", "coarsest", "edge_cut", "volume_before", "volume"), stats.tolist()))


def find_connected_components(nvar, factor, fmap):
    """``cc_id`` per variable -- the ``variable_to_cc`` table of messages.py:542-590."""
    return graph_order(nvar, factor, fmap, "components")[1]


def find_parts(nvar, factor, fmap, parts, method="auto"):
    """``(part id per variable, order)`` -- the ``variable_to_cc`` table of find_metis_parts (messages.py:593-670)
    for ``parts`` shards: variable ``order[i]`` belongs to the shard whose range holds ``i``.  ``method`` "auto":
    the order with the smallest communication volume among the caller's own ids (components together), the
    breadth-first walk and the multilevel partitioner -- never worse than what came in.  "multilevel" / "bfs" /
    "mas" / "median" / "components": that order alone."""
    n = int(nvar)
    if method == "auto":
        best = None
        for m in ("components", "bfs", "multilevel"):
            o = multilevel_order(n, factor, fmap, parts)[0] if m == "multilevel" else graph_order(n, factor, fmap, m)[0]
            vol = comm_volume(n, factor, fmap, parts, o)
            if best is None or vol < best[0]:
                best = (vol, o)
        order = best[1]
    elif method == "multilevel":
        order = multilevel_order(n, factor, fmap, parts)[0]
    else:
        order, _, _ = graph_order(nvar, factor, fmap, method)
    bounds = (np.arange(parts + 1, dtype=np.int64) * n) // parts
    part = np.empty(n, np.int64)
    part[order] = np.searchsorted(bounds, np.arange(n, dtype=np.int64), side="right") - 1
    return part, order


def comm_volume(nvar, factor, fmap, parts, order=None):
    """Values one exchange of the ``parts``-way range partition moves: (variable, foreign shard) pairs read across
    the cut (METIS' ``objtype = vol``).  ``order``: the partition cuts the ids of that order instead."""
    f, m = _records(factor, fmap)
    new_id = None
    if order is not None:
        new_id = np.empty(int(nvar), np.int64)
        new_id[np.asarray(order, np.int64)] = np.arange(int(nvar), dtype=np.int64)
    vol = C.c_int64()
    _lib.check(_lib.lib().nsk_comm_volume(int(nvar), len(f), _lib.ptr(f), len(m), _lib.ptr(m),
                                          _lib.ptr(new_id) if new_id is not None else C.c_void_p(0), int(parts), C.byref(vol)))
    return int(vol.value)


def relabel(graph, order):
    """The graph ``(weight, variable, factor, fmap, domain_mask, edges)`` with variable ``order[i]`` renamed ``i``.
    Factors keep their order and their members' order (so a variable's factor list keeps its order, and potentials
    their float64 sums); weights keep their ids.  Factors that look their head up at its literal edge index
    (inference.py:243 without ``head_by_vid``) do not survive a renaming: use ``head_by_vid``."""
    weight, variable, factor, fmap, domain_mask, edges = graph
    order = np.asarray(order, np.int64)
    n = len(variable)
    assert len(order) == n and np.array_equal(np.sort(order), np.arange(n)), "order must be a permutation of the variable ids"
    new_id = np.empty(n, np.int64)
    new_id[order] = np.arange(n, dtype=np.int64)
    fm = fmap.copy()
    fm["vid"] = new_id[fmap["vid"]]
    return weight, variable[order].copy(), factor, fm, np.ascontiguousarray(np.asarray(domain_mask)[order]), edges
