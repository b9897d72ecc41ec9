"""Two real processes on one GPU (gloo rendezvous, both ranks on cuda:0): the multi-rank host path
-- owned ranges, ghost needs gathered with torch.distributed, boundary planning, PartitionedSampler's
per-sweep exchange through torch collectives on the library's own device buffers, weight-delta
all-reduce -- against the oracle's emulation of the partitioned semantics.  (RCCL itself needs one
device per rank: the 1-rank RCCL tests in test_hip_parity.py and the driver's 8-GPU run cover it.)"""
import os
import sys

import pytest

from util import run_ranks

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["grid", "lr"])
@pytest.mark.parametrize("mode", ["gibbs", "learn"])
@pytest.mark.parametrize("layout", ["whole", "local", "p2p", "p2plocal"])
def test_two_ranks_one_gpu(kind, mode, layout):
    """layout "local": every rank holds only its shard of the graph -- owned variables, the ghosts
    they read, the factors that touch them (graphgen.extract_shard; what the reference's minions
    load, salt/src/numbskull_minion.py:185) -- and boundaries are planned in global ids.
    layout "p2p": the sweeps exchange boundaries by peer writes into hipIpc-mapped buffers with flags
    (nsk_gibbs_sweeps_p2p; learning: both chains and the epoch's weight deltas, nsk_learn_sweeps_p2p)
    instead of collectives; "p2plocal": the same on shard-local graphs (what bench.py --gpus N runs)."""
    r = run_ranks([os.path.join(HERE, "multirank_worker.py"), kind, mode, layout])
    if r.returncode != 0:                      # the workers' full output, for the session log
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/multirank_%s_%s_%s.log" % (kind, mode, layout), "w") as f:
            f.write(r.stdout + "\n----- stderr -----\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


@pytest.mark.parametrize("kind,mode", [("lr", "gibbs"), ("lr", "learn"), ("grid", "gibbs")])
def test_two_ranks_large_exchange_launches(kind, mode):
    """The many-block launches of a large exchange (k_p2p_push_big / k_p2p_wait / k_p2p_unpack_big; lists beyond 2^16
    values or 2^16 weights take them) on the small graphs: NSK_P2P_BIG_MIN=0 sends every exchange that way.  The grid
    run keeps the exchange kernels (NSK_NO_P2P_FUSE) and is long enough for captured sweep sequences: the launches then
    take their tags from the device counter."""
    r = run_ranks([os.path.join(HERE, "multirank_worker.py"), kind, mode, "p2plocal"],
                  env=dict(os.environ, NSK_DIAG="1", NSK_P2P_BIG_MIN="0", NSK_NO_P2P_FUSE="1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
