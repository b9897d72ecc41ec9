"""Shared helpers of the parity tests."""

import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numbskull_amd                                   # noqa: E402
from numbskull_amd import graphgen                     # noqa: E402
from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar   # noqa: E402
from oracle import binding as orc                      # noqa: E402


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def session(g, **kw):
    """NumbSkull session + its FactorGraph for a (weight, variable, factor, fmap, dm, edges) tuple."""
    kw.setdefault("quiet", True)
    ns = numbskull_amd.NumbSkull(**kw)
    w, v, f, fm, dm, edges = [x.copy() if isinstance(x, np.ndarray) else x for x in g]
    ns.loadFactorGraph(w, v, f, fm, dm, int(edges))
    return ns, ns.factorGraphs[0]


def oracle_of(fg, head_by_vid=False, layout=True):
    """Oracle view of the arrays a product FactorGraph holds.  Device mode keys its generator by
    the variables' positions in the library's compiled layout: on a GPU box
    the oracle is handed that layout (``layout=False`` for host-only uses) -- after checking, for
    EVERY graph the parity tests touch, what bit-equality with the oracle cannot show by itself:
    that the generator ids of the sampled variables are distinct (two variables sharing a Philox
    counter would still "equal the oracle") and that the colouring the library chose is valid."""
    og = orc.Graph(fg.weight, fg.variable, fg.factor, fg.fmap, fg.vmap, fg.factor_index,
                   head_by_vid=head_by_vid)
    if layout and numbskull_amd._lib.device_count() > 0:
        ids, color = fg.layout(), fg.colors()
        check_layout(ids, color)
        bad = og.check_coloring(color)
        assert bad == (-1, -1), ("variable reads a variable of its own colour", bad)
        gen = fg.generators()
        sampled = np.asarray(color) >= 0
        assert np.array_equal(gen[sampled] & 0xFFFFFFFFFF, np.asarray(ids)[sampled]) and (gen[~sampled] == -1).all()
        og.set_rng_ids(np.where(sampled, gen, ids))
        tag = fg.own_range[0] if fg.own_range is not None else 0
        if fg.global_ids is not None and fg.own_range is not None and len(fg.global_ids):     # shard-local graph
            tag = int(fg.global_ids[tag]) if tag < len(fg.global_ids) else int(fg.global_ids[-1]) + 1
        og.set_rng_tag(tag)
        info = fg.info()
        og.set_grad_shift(info["grad_shift"])
        og.device_lag = bool(info["learn_lag"])
    return og


def check_layout(ids, color):
    """Generator ids (= internal ids) must be injective: over the sampled variables (their Philox
    counters) and over all variables (their slots in the device's value arrays)."""
    ids = np.asarray(ids, np.int64)
    assert len(ids) == 0 or (ids.min() >= 0 and np.bincount(ids).max() == 1), \
        "two variables share an internal id"
    sampled = ids[np.asarray(color) >= 0]
    assert len(np.unique(sampled)) == len(sampled), "two sampled variables share a generator id"


def phases_from_colors(color):
    """Colour-major visiting order the oracle's device mode needs: (order, phase_start)."""
    color = np.asarray(color)
    sampled = np.nonzero(color >= 0)[0]
    order = sampled[np.argsort(color[sampled], kind="stable")]
    ncol = int(color.max()) + 1 if len(sampled) else 0
    phase_start = np.zeros(ncol + 1, np.int64)
    np.cumsum(np.bincount(color[sampled], minlength=ncol), out=phase_start[1:])
    return order.astype(np.int64), phase_start


def check_coloring(fg, color, head_by_vid=False):
    """No two sampled variables of one colour may share a factor (or be tied by the literal
    head index of IMPLY_MLN-type factors)."""
    f, fm = fg.factor, fg.fmap
    for fid in range(len(f)):
        s, a = int(f[fid]["ftv_offset"]), int(f[fid]["arity"])
        if int(f[fid]["factorFunction"]) == -1:        # NOOP reads no member: no constraint
            continue
        members = set(int(x) for x in fm["vid"][s:s + a])
        if int(f[fid]["factorFunction"]) in (13, 16, 17) and not head_by_vid:
            members.add(s + a - 1)
        cols = [color[m] for m in members if color[m] >= 0]
        assert len(cols) == len(set(cols)), ("colour clash in factor", fid)


def exact_marginals(og, weight_value, sample_mask=None):
    """Brute-force marginals of a tiny graph from the oracle's potentials: enumerates every
    assignment of the sampled variables (others fixed at initialValue)."""
    var = og.variable
    n = len(var)
    free = [i for i in range(n) if sample_mask is None or sample_mask[i]]
    cards = [int(var[i]["cardinality"]) for i in free]
    total = int(np.prod(cards))
    assert total <= 1 << 18
    base = var["initialValue"].astype(np.int64).copy()
    nf = len(og.factor)
    logp = np.zeros(total)
    states = np.zeros((total, len(free)), np.int64)
    for s in range(total):
        x, r = base.copy(), s
        for j, i in enumerate(free):
            x[i] = r % cards[j]
            states[s, j] = x[i]
            r //= cards[j]
        e = 0.0
        for fid in range(nf):
            rc, val = og.eval_factor(fid, -1, 0, x)
            assert rc == 0
            e += weight_value[int(og.factor[fid]["weightId"])] * val
        logp[s] = e
    p = np.exp(logp - logp.max())
    p /= p.sum()
    out = {}
    for j, i in enumerate(free):
        out[i] = np.array([p[states[:, j] == k].sum() for k in range(cards[j])])
    return out


def free_port():
    """A TCP port for a rendezvous on 127.0.0.1, free now and OUTSIDE the kernel's ephemeral range (32768-60999):
    a port probed with bind(0) can be handed to somebody's outgoing connection before the rendezvous server binds it
    (seen once in 339 GPU tests: EADDRINUSE)."""
    import random
    import socket
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free port in 20000-31999")


def run_ranks(script_args, nproc=2, env=None, timeout=600, tries=3):
    """`python -m torch.distributed.run --nproc-per-node nproc script_args...` on a fresh port; a rendezvous that
    loses its port to another process all the same is started again (up to `tries` times).  Returns the
    CompletedProcess of the last attempt."""
    import subprocess
    r = None
    for _ in range(tries):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + list(script_args)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    return r
