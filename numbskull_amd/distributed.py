"""Range-partitioned sampling over several GPUs of one node (one process per GPU).

The reference scales out by giving every machine the full variable arrays, flagging the
variables a machine does not own with ``isEvidence == 4`` so the samplers skip them
(inference.py:21-23, learning.py:24-26; set at salt/src/numbskull_master.py:343 and
numbskull_minion.py:185), and copying owners' values to the replicas once per epoch over
Salt/TCP (numbskull_master.py:165-224); learned weights are merged as ``w += sum of deltas``
(numbskull_master.py:223-224, numbskull_minion.py:270-279).

Here: rank g owns variables ``[g*n//G, (g+1)*n//G)`` -- the reference's shard formula
(inference.py:17-18).  After every sweep only the **boundary** values travel: each rank packs the
owned variables some other rank reads into a small send buffer, the buffers are all-gathered over
RCCL/xGMI, and every rank scatters what it received into its value array.  Ghost values are
therefore one sweep old inside a sweep, exactly the reference's distributed semantics.  In
learning the evidence-chain values are exchanged too and the weight deltas of the epoch are summed
with an all-reduce.

Drivers for the per-sweep loop:
  * peer-to-peer (default on GPUs when its set-up self-test passes on every rank; NSK_P2P=0 or
    ``p2p=False`` switches it off): boundary lists are PAIRWISE (rank s sends rank d exactly what d reads
    from s); every rank writes its boundary values -- in learning also the evidence-chain values and the
    epoch's weight deltas -- straight into buffers of its peers (device memory mapped with hipIpc; xGMI
    between GPUs) and raises a flag there; a rank waits for the flags of its peers and scatters their
    values (nsk_gibbs_sweeps_p2p / nsk_learn_sweeps_p2p) -- no collective, no host round trip per sweep.
    The weight deltas of a learning epoch are merged by reduce-scatter + all-gather over the same buffers
    (rank q owns a slice of the weight vector, adds the ranks' deltas in rank order and hands the merged
    slice to everyone: bit-identical weights on all ranks).
    A peer whose flag does not arrive within NSK_P2P_TIMEOUT_S (default 30 s) is reported by
    ``check()``.  Exercised with several ranks sharing ONE device only (tests/); across devices it
    relies on hipIpc peer mappings and system-scope flags over xGMI;
  * native: the whole loop -- sweep kernels, pack, ncclAllGather, unpack (+ ncclAllReduce of the weight
    deltas) -- is enqueued from C++ on one stream (nsk_*_sweeps_exchange), the communicator being
    created from a ncclUniqueId that rank 0 makes and ``torch.distributed`` broadcasts;
  * torch: ``torch.distributed.all_gather_into_tensor`` on the library's staging buffers wrapped
    as tensors; used by the CPU (gloo) tests and as the last fallback.
"""

import ctypes as C
import sys
import os

import numpy as np


def shard_range(rank, world, nvar):
    """[start, end) of inference.py:17-18."""
    return (rank * nvar) // world, ((rank + 1) * nvar) // world


def plan_boundaries(needs_per_rank, world, nvar):
    """Boundary lists every rank agrees on.

    ``needs_per_rank[r]``: sorted ids rank r reads but does not own.  Returns ``(lists, slot)``:
    ``lists[src]`` = sorted ids owned by ``src`` that at least one other rank reads, and the common
    slot size (max length) of the gathered buffer."""
    allneed = np.unique(np.concatenate([np.asarray(n, np.int64) for n in needs_per_rank]
                                       + [np.empty(0, np.int64)]))
    lists = []
    for src in range(world):
        lo, hi = shard_range(src, world, nvar)
        lists.append(allneed[(allneed >= lo) & (allneed < hi)].astype(np.int32))
    slot = max([len(b) for b in lists] + [0])
    return lists, slot


def plan_pairs(needs_per_rank, world, nvar):
    """Pairwise boundary lists: ``pairs[s][d]`` = sorted ids owned by ``s`` that rank ``d`` reads."""
    bounds = np.array([shard_range(r, world, nvar)[0] for r in range(world)] + [nvar], np.int64)
    pairs = [[None] * world for _ in range(world)]
    for d in range(world):
        need = np.asarray(needs_per_rank[d], np.int64)
        cut = np.searchsorted(need, bounds)
        for s_ in range(world):
            pairs[s_][d] = need[cut[s_]:cut[s_ + 1]] if s_ != d else need[:0]
    return pairs


def gather_needs(dist, torch, needs, world, device):
    """All-gather the (ragged) need lists of all ranks."""
    if world == 1:
        return [np.asarray(needs, np.int32)]
    n = torch.tensor([len(needs)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes + [1])
    mine = torch.full((width,), -1, dtype=torch.int32, device=device)
    mine[:len(needs)] = torch.as_tensor(np.asarray(needs, np.int32), device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def merge_weight_deltas(dist, weights, start, group=None):
    """w = w_start + sum_g (w_g - w_start): the master's merge rule (numbskull_master.py:223-224)."""
    delta = weights - start
    dist.all_reduce(delta, group=group)
    weights.copy_(start + delta)


def weight_slice(q, world, nw):
    """First weight of rank q's slice in the sliced merge: the device's nsk_p2p_slice_lo (q * nw // world)."""
    return (int(q) * int(nw)) // int(world)


def merge_weight_deltas_sliced(dist, weights, start, rank, world, group=None):
    """The same merge as reduce-scatter + all-gather, the way the peer-to-peer path does it on the device
    (nsk_kernels_misc.h p2p_push / p2p_reduce_slice): rank q owns weights [q nw // W, (q + 1) nw // W), adds the W ranks'
    deltas of its slice IN RANK ORDER to w_start and hands the merged slice to everybody -- every rank ends with
    bit-identical weights whatever the collective library's own summation order, and only nw / W weights per rank
    cross the wire twice.  Slices of unequal length (nw not a multiple of W) are padded to the longest for the
    collectives."""
    torch = __import__("torch")
    nw = int(weights.numel())
    lo = [weight_slice(q, world, nw) for q in range(world + 1)]
    smax = max(lo[q + 1] - lo[q] for q in range(world))
    delta = weights - start
    # scatter: rank r sends slice q of its deltas to rank q (padded rows)
    send = [torch.zeros(smax, dtype=weights.dtype, device=weights.device) for _ in range(world)]
    for q in range(world):
        send[q][:lo[q + 1] - lo[q]] = delta[lo[q]:lo[q + 1]]
    recv = [torch.zeros(smax, dtype=weights.dtype, device=weights.device) for _ in range(world)]
    dist.all_to_all(recv, send, group=group) if dist.get_backend(group) != "gloo" else _all_to_all_by_gather(dist, torch, recv, send, rank, world, group)
    n_me = lo[rank + 1] - lo[rank]
    t = recv[0][:n_me].clone()
    for r in range(1, world):                       # rank order, like the device's loop
        t += recv[r][:n_me]
    mine = torch.zeros(smax, dtype=weights.dtype, device=weights.device)
    mine[:n_me] = start[lo[rank]:lo[rank + 1]] + t
    out = [torch.zeros(smax, dtype=weights.dtype, device=weights.device) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    for q in range(world):
        weights[lo[q]:lo[q + 1]] = out[q][:lo[q + 1] - lo[q]]


def _all_to_all_by_gather(dist, torch, recv, send, rank, world, group):
    """all_to_all for backends without it (gloo): W all-gathers of the padded rows."""
    for q in range(world):
        rows = [torch.zeros_like(send[q]) for _ in range(world)]
        dist.all_gather(rows, send[q], group=group)
        if q == rank:
            for r in range(world):
                recv[r].copy_(rows[r])


class _Watchdog(object):
    """Time box around one rung of the exchange set-up: a native call that never returns (a peer that died inside a
    collective, an IPC import that blocks) cannot be interrupted from Python, so after `seconds` the rank says where it
    hangs and exits with code 3 -- bench.py's ranks are CHILD processes of a parent that has not touched the GPU
    (spawn_ranks), which relays the code; nothing re-execs."""

    def __init__(self, seconds, what, ladder):
        self.seconds, self.what, self.ladder, self.t = seconds, what, ladder, None

    def __enter__(self):
        import threading

        def fire():
            print("[numbskull_amd] rank %s: %s did not finish within %.0f s (NSK_RUNG_TIMEOUT_S); ladder so far: %s"
                  % (self.ladder.get("rank"), self.what, self.seconds, self.ladder.get("tried")), file=sys.stderr, flush=True)
            os._exit(3)
        if self.seconds > 0:
            self.t = threading.Timer(self.seconds, fire)
            self.t.daemon = True
            self.t.start()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.t.cancel()
        return False


class _DevicePointer(object):
    """Expose a raw device allocation of the library to torch via __cuda_array_interface__."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class PartitionedSampler(object):
    """One rank's share of a range-partitioned graph.

    ``fg`` is a FactorGraph created with ``own_range=shard_range(rank, world, nvar)``.  The
    library's buffers are wrapped as torch tensors (no copies) and the library is pointed at
    torch's current stream, so sweeps and collectives are ordered by the stream.
    """

    def __init__(self, fg, dist, torch, rank, world, native=True, nvar_global=None, p2p=None, pf=None):
        """``pf``: the partial factors of a shard rewritten by ``graphgen.partial_factors`` (its fourth return value):
        aggregates over foreign members that this shard READS -- their owners compute and ship them (peer-to-peer
        path only; SURVEY.md section 8 f3, salt/src/messages.py:1333-1355)."""
        from . import _lib
        self.fg, self.dist, self.torch, self.rank, self.world = fg, dist, torch, rank, world
        self.pf = list(pf) if pf else []
        self.all_pf = None              # every rank's requests (encoded), filled by setup_exchange
        self.L = _lib.lib()
        self._lib = _lib
        h = fg._engine()
        self.h = h
        self.nvar = fg.variable.shape[0]
        # a shard-local graph (graphgen.extract_shard) holds its own variables and the ghosts it reads
        # under local ids: boundaries are planned in GLOBAL ids and translated for the library
        self.gids = fg.global_ids
        self.nvar_global = int(nvar_global) if nvar_global is not None else self.nvar
        info = fg.info()
        self.typestr = {1: "|i1", 4: "<i4"}[info["value_bytes"]]
        self.dev = "cuda:%d" % fg.device
        _lib.check(self.L.nsk_set_stream(h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        # the library keeps values in its own (layout) order: index tensor = internal id of every
        # variable; ``val`` / ``val_evid`` below are by-variable-id copies gathered through it
        iid = np.zeros(self.nvar, np.int32)
        nid = C.c_int64()
        _lib.check(self.L.nsk_graph_get_layout(h, _lib.ptr(iid), C.byref(nid)))
        self.iid = torch.as_tensor(iid.astype(np.int64), device=self.dev)
        self.nid = int(nid.value)
        self._wrap_values()
        self.w = self._wrap(_lib.BUF_WEIGHT, fg.weight.shape[0], "<f8")
        self.native = False
        self.p2p = False
        self.lists, self.slot = None, 0
        self.all_needs = None
        # which rung of the exchange ladder this rank ends on, and how it got there (bench.py prints it per rank):
        # fused (inside the table launches) -> peer-to-peer exchange kernels -> native RCCL loop -> torch.distributed
        self.ladder = {"rank": rank, "tried": [], "seconds": {}, "rung": "single rank" if world == 1 else None}
        if world > 1:
            import time
            box = float(os.environ.get("NSK_RUNG_TIMEOUT_S", "180"))      # a rung that hangs ends the rank, loudly
            t0 = time.time()
            with _Watchdog(box, "setting up the collective exchange (boundary lists, RCCL communicator)", self.ladder):
                self.setup_exchange(native)
            self.ladder["seconds"]["collective set-up"] = round(time.time() - t0, 3)
            self.ladder["tried"].append("native RCCL loop: %s" % ("ready" if self.native else "not available (torch.distributed loop)"))
            # peer-to-peer exchange for inference and learning sweeps (NSK_P2P=0 or p2p=False: off).  Set-up
            # ends with a self-test -- two real exchanges, one per buffer parity -- and every rank must
            # pass or none uses it (the collective loop stays as the fallback)
            if p2p if p2p is not None else os.environ.get("NSK_P2P", "1") != "0":
                t0 = time.time()
                with _Watchdog(box, "setting up the peer-to-peer exchange (hipIpc mappings, self-tests)", self.ladder):
                    self.p2p = self._init_p2p()
                self.ladder["seconds"]["peer-to-peer set-up + self-tests"] = round(time.time() - t0, 3)
                self.ladder["tried"].append("peer-to-peer self-test: %s" % ("passed on every rank" if self.p2p else "failed on some rank"))
                if self.p2p:
                    fused = bool(fg.info()["p2p_fused"])
                    self.ladder["tried"].append("fused exchange: %s" % ("planned and self-tested on every rank" if fused else
                                                                        "not for this handle (not table launches only, or its self-test failed somewhere)"))
            else:
                self.ladder["tried"].append("peer-to-peer: switched off")
            self.ladder["rung"] = ("fused into the table launches" if self.p2p and fg.info()["p2p_fused"] else
                                   "peer-to-peer exchange kernels" if self.p2p else
                                   "native RCCL loop" if self.native else "torch.distributed loop")
            if self.pf and not self.p2p:
                raise RuntimeError("partial factors need the peer-to-peer exchange (their aggregates are computed by the "
                                   "exchange kernels); it could not be set up")

    @property
    def val(self):
        """var_value by variable id (a copy gathered from the device buffer's internal order)."""
        return self.val_raw[self.iid] if self.nvar else self.val_raw

    @property
    def val_evid(self):
        return self.val_evid_raw[self.iid] if self.nvar else self.val_evid_raw

    def _wrap_values(self):
        """(Again after nsk_pf_setup: the value arrays move when they grow by the aggregates' slots.)"""
        self.val_raw = self._wrap(self._lib.BUF_VALUE, self.nid, self.typestr)
        self.val_evid_raw = self._wrap(self._lib.BUF_VALUE_EVID, self.nid, self.typestr)

    def _wrap(self, which, nelem, ts):
        p, nb = C.c_void_p(), C.c_int64()
        self._lib.check(self.L.nsk_device_buffer(self.h, which, C.byref(p), C.byref(nb)))
        if nelem == 0 or not p.value:
            return self.torch.empty(0, device=self.dev)
        return self.torch.as_tensor(_DevicePointer(p.value, nelem, ts), device=self.dev)

    def install_boundaries(self, lists, slot):
        """Hand the agreed boundary lists to the library and wrap its staging buffers."""
        _lib = self._lib
        self.lists, self.slot = lists, int(slot)
        send = np.ascontiguousarray(self._local(lists[self.rank]), np.int32)
        assert (send >= 0).all(), "a boundary list names a variable this shard does not hold"
        recv = np.ascontiguousarray(self._local(np.concatenate(lists + [np.empty(0, np.int32)])), np.int32)
        off = np.zeros(self.world + 1, np.int64)
        np.cumsum([len(b) for b in lists], out=off[1:])
        _lib.check(self.L.nsk_exchange_setup(self.h, self.world, self.rank, _lib.ptr(send), len(send),
                                             _lib.ptr(recv), _lib.ptr(off), self.slot))
        self.send = self._wrap(_lib.BUF_SEND, self.slot, self.typestr)
        self.recv = self._wrap(_lib.BUF_RECV, self.slot * self.world, self.typestr)
        self.send_evid = self._wrap(_lib.BUF_SEND_EVID, self.slot, self.typestr)
        self.recv_evid = self._wrap(_lib.BUF_RECV_EVID, self.slot * self.world, self.typestr)

    def _local(self, ids):
        """Global variable ids -> this handle's ids (-1: not held here); identity for a whole-graph handle."""
        ids = np.asarray(ids, np.int64)
        if self.gids is None:
            return ids
        at = np.minimum(np.searchsorted(self.gids, ids), len(self.gids) - 1) if len(self.gids) else np.zeros(len(ids), np.int64)
        return np.where(len(self.gids) and self.gids[at] == ids, at, -1) if len(ids) else ids

    def global_needs(self):
        """Sorted GLOBAL ids of the variables this shard reads but does not own (the ghosts that stand for partial-
        factor aggregates -- synthetic ids from ``nvar_global`` on -- are not variables of any rank: ``pf_requests``)."""
        needs = self.fg.ghost_needs()
        if self.gids is None:
            return needs
        g = self.gids[needs]
        return g[g < self.nvar_global].astype(np.int32)

    def pf_requests(self):
        """This shard's partial factors as one int32 array: per aggregate ``owner, op, count, member global ids...``."""
        out = []
        for q, op, members, _ in self.pf:
            out += [int(q), int(op), len(members)] + [int(m) for m in members]
        return np.asarray(out, np.int32)

    @staticmethod
    def _decode_pf(arr):
        arr = np.asarray(arr, np.int64)
        out, i = [], 0
        while i < len(arr):
            q, op, c = int(arr[i]), int(arr[i + 1]), int(arr[i + 2])
            out.append((q, op, arr[i + 3:i + 3 + c].copy()))
            i += 3 + c
        return out

    def setup_exchange(self, native=True):
        needs = self.global_needs()
        self.all_needs = gather_needs(self.dist, self.torch, needs, self.world, self.dev)
        self.all_pf = gather_needs(self.dist, self.torch, self.pf_requests(), self.world, self.dev)
        lists, slot = plan_boundaries(self.all_needs, self.world, self.nvar_global)
        self.install_boundaries(lists, slot)
        if native:
            self.native = self._init_native()

    def _init_native(self):
        """ncclUniqueId from rank 0, broadcast with torch.distributed, ncclCommInitRank per rank."""
        torch, dist = self.torch, self.dist
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = path if os.path.exists(path) else ""
        uid = (C.c_uint8 * 128)()
        ok = 1
        if self.rank == 0:
            ok = int(self.L.nsk_comm_unique_id(path.encode(), uid) == 0)
        t = torch.tensor(list(bytes(uid)) + [ok], dtype=torch.uint8, device=self.dev)
        if self.world > 1:
            dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tolist())
        if raw[128] != 1:
            return False
        buf = (C.c_uint8 * 128).from_buffer_copy(raw[:128])
        rc = self.L.nsk_comm_init(self.h, self.world, self.rank, buf, path.encode())
        flag = torch.tensor([int(rc == 0)], dtype=torch.int32, device=self.dev)
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # native only if it works everywhere
        return bool(flag.item())

    def p2p_lists(self):
        """Arguments of nsk_p2p_setup for this rank, from the gathered need lists: what every peer reads
        from this rank, what this rank reads from every peer (local ids), and where this rank's segment
        lies in every peer's receive list."""
        pairs = plan_pairs(self.all_needs, self.world, self.nvar_global)
        me, W = self.rank, self.world
        # partial factors: reader d's requests, in d's order, follow the plain values of every (owner, d) segment
        reqs = [self._decode_pf(a) for a in (self.all_pf if self.all_pf is not None else [np.empty(0, np.int32)] * W)]
        npf = [[sum(1 for q, _, _ in reqs[d] if q == s_) for d in range(W)] for s_ in range(W)]      # [owner][reader]
        self.pf_out = []                # the aggregates this rank computes: (op, member global ids), slot = index
        out_slots = [[] for _ in range(W)]
        for d in range(W):
            for q, op, members in reqs[d]:
                if q == me:
                    out_slots[d].append(len(self.pf_out))
                    self.pf_out.append((op, members))
        nloc = self.nvar
        send = [np.concatenate([self._local(pairs[me][q]), nloc + np.asarray(out_slots[q], np.int64)]) for q in range(W)]
        mine = [(q, lid) for q, _, _, lid in self.pf]
        recv = [np.concatenate([self._local(pairs[q][me]), np.asarray([lid for o, lid in mine if o == q], np.int64)]) for q in range(W)]
        soff, roff = np.zeros(W + 1, np.int64), np.zeros(W + 1, np.int64)
        np.cumsum([len(x) for x in send], out=soff[1:])
        np.cumsum([len(x) for x in recv], out=roff[1:])
        cnt = lambda s_, q: len(pairs[s_][q]) + npf[s_][q]
        base = np.array([sum(cnt(s_, q) for s_ in range(me)) for q in range(W)], np.int64)
        total = np.array([sum(cnt(s_, q) for s_ in range(W)) for q in range(W)], np.int64)
        cat = lambda xs: np.ascontiguousarray(np.concatenate(xs + [np.empty(0, np.int64)]), np.int32)
        return cat(send), soff, cat(recv), roff, base, total

    def p2p_setup(self):
        _lib = self._lib
        send, soff, recv, roff, base, total = self.p2p_lists()
        assert (send >= 0).all() and (recv >= 0).all(), "a boundary list names a variable this shard does not hold"
        if self.pf_out:                 # the aggregates this rank's readers asked for (before the lists that name them)
            ops = np.asarray([op for op, _ in self.pf_out], np.uint8)
            moff = np.zeros(len(self.pf_out) + 1, np.int64)
            np.cumsum([len(m) for _, m in self.pf_out], out=moff[1:])
            mem = np.ascontiguousarray(self._local(np.concatenate([m for _, m in self.pf_out])), np.int32)
            assert (mem >= 0).all(), "a partial factor names a variable this shard does not hold"
            rc = self.L.nsk_pf_setup(self.h, len(ops), _lib.ptr(ops), _lib.ptr(moff), _lib.ptr(mem))
            if rc:
                return rc
            self._wrap_values()
        return self.L.nsk_p2p_setup(self.h, self.world, self.rank, _lib.ptr(send), _lib.ptr(soff), _lib.ptr(recv),
                                    _lib.ptr(roff), _lib.ptr(base), _lib.ptr(total))

    def _init_p2p(self):
        """Pairwise lists to the library, then the hipIpc handle of every rank's buffer, all-gathered with
        torch.distributed; each rank maps its peers' (nsk_p2p_import)."""
        torch, dist = self.torch, self.dist

        def agreed(ok):
            flag = torch.tensor([int(bool(ok))], dtype=torch.int32, device=self.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
        if not agreed(self.p2p_setup() == 0):
            return False
        mine = (C.c_uint8 * 64)()
        ok = int(self.L.nsk_p2p_export(self.h, mine, None) == 0)
        t = torch.tensor(list(bytes(mine)) + [ok], dtype=torch.uint8, device=self.dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t)
        raw = [bytes(o.cpu().numpy().tolist()) for o in outs]
        if not all(r[64] == 1 for r in raw):
            return False
        table = (C.c_uint8 * (64 * self.world)).from_buffer_copy(b"".join(r[:64] for r in raw))
        if not agreed(self.L.nsk_p2p_import(self.h, table) == 0):
            return False
        # self-test: two learning-mode exchanges -- the very kernels, mappings and flags of the sweep loops,
        # one exchange per buffer parity -- whose payload is a pattern that depends on sender, element,
        # exchange and chain, compared on the receiving side (values, weight-delta slices and merged
        # weights): a peer's WRITES must be visible here, not only its flags.  State and weights untouched
        def why():
            msg = self._lib.lib().nsk_last_error()
            return msg.decode() if isinstance(msg, bytes) else str(msg)

        def reset_all():
            # a failed self-test leaves tags advanced and patterns in the receive blocks of the ranks that passed:
            # every rank starts again from a clean allocation (nobody still writes / nobody starts early: barriers)
            dist.barrier()
            self.L.nsk_p2p_reset(self.h)
            dist.barrier()
        rc = 0
        for _ in range(2):
            rc = rc or self.L.nsk_p2p_selftest(self.h, 1, 0)
        rc = rc or self.L.nsk_p2p_check(self.h)
        if rc:
            print("[numbskull_amd] rank %d: peer-to-peer self-test failed (%s): every rank takes the collective exchange"
                  % (self.rank, why()), file=sys.stderr, flush=True)
        if not agreed(rc == 0):
            reset_all()
            return False
        # shards that live in table segments exchange INSIDE their class launches (nsk_graph_info.p2p_fused): system-
        # coherent loads / stores of peer memory, no fences.  That protocol gets a self-test of its own, and every
        # rank keeps the exchange kernels unless every rank passes it
        fused = bool(self.fg.info()["p2p_fused"])
        rc = 0
        if fused:
            for _ in range(2):
                rc = rc or self.L.nsk_p2p_selftest(self.h, 2, 0)
            rc = rc or self.L.nsk_p2p_check(self.h)
            if rc:
                print("[numbskull_amd] rank %d: self-test of the fused exchange failed (%s): every rank keeps the exchange kernels"
                      % (self.rank, why()), file=sys.stderr, flush=True)
        any_fused = not agreed(not fused)
        if not agreed(fused and rc == 0):
            self.L.nsk_p2p_fuse(self.h, 0)
            if any_fused:                       # (the fused self-test ran on some rank: its tags and the peers' blocks are used)
                reset_all()
        return True

    def check(self):
        """Raise if a peer-to-peer exchange since the last check timed out (synchronises the stream)."""
        if self.p2p:
            self._lib.check(self.L.nsk_p2p_check(self.h))

    # ------------------------------------------------------------------ per-sweep loops
    def _exchange(self, which, send, recv):
        self._lib.check(self.L.nsk_exchange_pack(self.h, which))
        if self.slot > 0:
            self.dist.all_gather_into_tensor(recv, send)
        self._lib.check(self.L.nsk_exchange_unpack(self.h, which))

    def gibbs(self, nsweeps, sample_evidence=True, burnin=False):
        _lib = self._lib
        if self.world == 1:
            _lib.check(self.L.nsk_gibbs_sweeps(self.h, nsweeps, int(sample_evidence), int(burnin)))
        elif self.p2p:
            _lib.check(self.L.nsk_gibbs_sweeps_p2p(self.h, nsweeps, int(sample_evidence), int(burnin)))
            self.check()        # (one stream synchronisation per call: a timed-out peer raises here, not later)
        elif self.native:
            _lib.check(self.L.nsk_gibbs_sweeps_exchange(self.h, nsweeps, int(sample_evidence), int(burnin)))
        else:
            for _ in range(nsweeps):
                _lib.check(self.L.nsk_gibbs_sweeps(self.h, 1, int(sample_evidence), int(burnin)))
                self._exchange(_lib.BUF_VALUE, self.send, self.recv)

    def phase_timings(self, nsweeps=20, sample_evidence=True, learn=None):
        """Diagnostic: mean microseconds of the phases of one sweep of this shard, each bracketed by
        HIP events on the library's stream and issued on its own (so every figure carries one launch
        latency): the sweep kernels, then the exchange -- peer-to-peer push + wait/unpack (learning:
        both chains + weight deltas + merge), or pack / collective / unpack.  ``learn``: (step,
        regularization, reg_param, truncation) for learning sweeps (peer-to-peer path only)."""
        _lib, L, h = self._lib, self.L, self.h
        if learn is not None and not self.p2p:
            return {}
        ms, nl = C.c_double(), C.c_int64()

        def timed(fn):
            _lib.check(L.nsk_profile_begin(h))
            fn()
            _lib.check(L.nsk_profile_end(h, C.byref(ms), C.byref(nl)))
            return ms.value * 1e3
        out = {"sweep": 0.0}
        for _ in range(nsweeps):
            if learn is not None:
                out["sweep"] += timed(lambda: _lib.check(L.nsk_learn_sweeps(h, 1, float(learn[0]), 1.0, int(learn[1]),
                                                                                float(learn[2]), int(learn[3]), 0)))
            else:
                out["sweep"] += timed(lambda: _lib.check(L.nsk_gibbs_sweeps(h, 1, int(sample_evidence), 1)))
            if self.world == 1:
                continue
            if self.p2p:
                out["p2p_exchange"] = out.get("p2p_exchange", 0.0) + timed(
                    lambda: _lib.check(L.nsk_p2p_exchange(h, int(learn is not None), 0)))
            else:
                out["pack"] = out.get("pack", 0.0) + timed(lambda: _lib.check(L.nsk_exchange_pack(h, _lib.BUF_VALUE)))
                t0 = self.torch.cuda.Event(enable_timing=True)
                t1 = self.torch.cuda.Event(enable_timing=True)
                t0.record()
                if self.slot > 0:
                    self.dist.all_gather_into_tensor(self.recv, self.send)
                t1.record()
                t1.synchronize()
                out["all_gather"] = out.get("all_gather", 0.0) + t0.elapsed_time(t1) * 1e3
                out["unpack"] = out.get("unpack", 0.0) + timed(lambda: _lib.check(L.nsk_exchange_unpack(h, _lib.BUF_VALUE)))
        return {k: v / nsweeps for k, v in out.items()}

    def learn(self, nsweeps, step, decay, regularization, reg_param, truncation,
              learn_non_evidence=False):
        _lib = self._lib
        args = (int(regularization), float(reg_param), int(truncation), int(learn_non_evidence))
        if self.world == 1:
            _lib.check(self.L.nsk_learn_sweeps(self.h, nsweeps, float(step), float(decay), *args))
        elif self.p2p:
            _lib.check(self.L.nsk_learn_sweeps_p2p(self.h, nsweeps, float(step), float(decay), *args))
            self.check()
        elif self.native:
            _lib.check(self.L.nsk_learn_sweeps_exchange(self.h, nsweeps, float(step), float(decay), *args))
        else:
            for _ in range(nsweeps):
                start = self.w.clone()
                _lib.check(self.L.nsk_learn_sweeps(self.h, 1, float(step), 1.0, *args))
                self._exchange(_lib.BUF_VALUE, self.send, self.recv)
                self._exchange(_lib.BUF_VALUE_EVID, self.send_evid, self.recv_evid)
                merge_weight_deltas(self.dist, self.w, start)
                step *= decay
