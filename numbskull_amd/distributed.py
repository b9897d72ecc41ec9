"""Range-partitioned sampling over several GPUs of one node (one process per GPU).

The reference scales out by giving every machine the full variable arrays, flagging the
variables a machine does not own with ``isEvidence == 4`` so the samplers skip them
(inference.py:21-23, learning.py:24-26; set at salt/src/numbskull_master.py:343 and
numbskull_minion.py:185), and copying owners' values to the replicas once per epoch over
Salt/TCP (numbskull_master.py:165-224); learned weights are merged as ``w += sum of deltas``
(numbskull_master.py:223-224, numbskull_minion.py:270-279).

Here: rank g owns variables ``[g*n//G, (g+1)*n//G)`` -- the reference's shard formula
(inference.py:17-18) -- and after every sweep the owned slices of the value array are
all-gathered with ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Ghost values are therefore one sweep old inside a sweep, exactly the reference's
distributed semantics.  In learning the evidence-chain values are exchanged too and the weight
deltas of the epoch are summed with an all-reduce.
"""

import numpy as np


def shard_range(rank, world, nvar):
    """[start, end) of inference.py:17-18."""
    return (rank * nvar) // world, ((rank + 1) * nvar) // world


def exchange_values(dist, tensor, world, nvar, group=None):
    """All-gather the owned slices of a per-variable tensor in place.  Equal shards use one
    all_gather_into_tensor; ragged shards fall back to one broadcast per owner."""
    bounds = [shard_range(r, world, nvar) for r in range(world)]
    sizes = {e - b for b, e in bounds}
    rank = dist.get_rank(group)
    if len(sizes) == 1 and bounds[-1][1] == nvar and tensor.is_contiguous():
        b, e = bounds[rank]
        dist.all_gather_into_tensor(tensor[:nvar], tensor[b:e], group=group)
        return
    for r, (b, e) in enumerate(bounds):
        if e > b:
            dist.broadcast(tensor[b:e], src=r, group=group)


def merge_weight_deltas(dist, weights, start, group=None):
    """w = w_start + sum_g (w_g - w_start): the master's merge rule (numbskull_master.py:223-224)."""
    delta = weights - start
    dist.all_reduce(delta, group=group)
    weights.copy_(start + delta)


class _DevicePointer(object):
    """Expose a raw device allocation of the library to torch via __cuda_array_interface__."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class PartitionedSampler(object):
    """One rank's share of a range-partitioned graph.

    ``fg`` is a FactorGraph created with ``own_range=shard_range(rank, world, nvar)``.  The
    library's value buffers are wrapped as torch tensors (no copies) and the library is pointed
    at torch's current stream, so sweeps and collectives are ordered by the stream.
    """

    def __init__(self, fg, dist, torch, rank, world):
        import ctypes as C
        from . import _lib
        self.fg, self.dist, self.torch, self.rank, self.world = fg, dist, torch, rank, world
        self.L = _lib.lib()
        self._lib = _lib
        h = fg._engine()
        self.h = h
        self.nvar = fg.variable.shape[0]
        info = fg.info()
        typestr = {1: "|i1", 4: "<i4"}[info["value_bytes"]]
        dev = "cuda:%d" % fg.device
        _lib.check(self.L.nsk_set_stream(h, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

        def wrap(which, nelem, ts):
            p, nb = C.c_void_p(), C.c_int64()
            _lib.check(self.L.nsk_device_buffer(h, which, C.byref(p), C.byref(nb)))
            return torch.as_tensor(_DevicePointer(p.value, nelem, ts), device=dev)

        self.val = wrap(_lib.BUF_VALUE, self.nvar, typestr)
        self.val_evid = wrap(_lib.BUF_VALUE_EVID, self.nvar, typestr)
        self.w = wrap(_lib.BUF_WEIGHT, fg.weight.shape[0], "<f8")

    def gibbs(self, nsweeps, sample_evidence=True, burnin=False):
        for _ in range(nsweeps):
            self._lib.check(self.L.nsk_gibbs_sweeps(self.h, 1, int(sample_evidence), int(burnin)))
            if self.world > 1:
                exchange_values(self.dist, self.val, self.world, self.nvar)

    def learn(self, nsweeps, step, decay, regularization, reg_param, truncation,
              learn_non_evidence=False):
        for _ in range(nsweeps):
            start = self.w.clone()
            self._lib.check(self.L.nsk_learn_sweeps(self.h, 1, float(step), 1.0,
                                                    int(regularization), float(reg_param),
                                                    int(truncation), int(learn_non_evidence)))
            if self.world > 1:
                exchange_values(self.dist, self.val, self.world, self.nvar)
                exchange_values(self.dist, self.val_evid, self.world, self.nvar)
                merge_weight_deltas(self.dist, self.w, start)
            step *= decay
