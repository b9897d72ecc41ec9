"""Boundary size of the reference's range partition on the config-#5 generator (SURVEY.md section
8(f) rank 4 asks whether a graph-aware partitioner is needed): for G shards, how many of a shard's
variables are read by another shard, and how many foreign variables a shard reads.  Host-only
(nsk_graph_plan_needs); no GPU."""
import io, sys, os, time
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import numbskull_amd
from numbskull_amd import graphgen
from numbskull_amd.distributed import shard_range, plan_boundaries

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for gf in (0.01, 0.0):
    g = graphgen.mixed_lr_graph(n, seed=20240603, global_frac=gf)
    needs = []
    for r in range(world):
        ns = numbskull_amd.NumbSkull(quiet=True, head_by_vid=True)
        with redirect_stdout(io.StringIO()):
            ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                               own_range=shard_range(r, world, n))
        needs.append(ns.factorGraphs[0].ghost_needs(host_only=True))
    lists, slot = plan_boundaries(needs, world, n)
    own = n / world
    print("global_frac %.2f: %d variables, %d shards: ghosts read per shard mean %.0f (%.2f%% of owned), "
          "boundary (owned, read elsewhere) mean %.0f (%.2f%% of owned), slot %d"
          % (gf, n, world, np.mean([len(x) for x in needs]), 100 * np.mean([len(x) for x in needs]) / own,
             np.mean([len(x) for x in lists]), 100 * np.mean([len(x) for x in lists]) / own, slot))
