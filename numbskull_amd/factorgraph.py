"""FactorGraph: owner of the graph/state arrays and driver of burn-in, inference and learning.

Keeps the public face of the reference class (numbskull/factorgraph.py:27-229: constructor
signature, attributes, method names, text dumps) so that callers -- the CLI, the reference's
smoke scripts, code that patches ``var_value`` / ``weight_value`` between epochs -- keep working.
What changed is what happens below ``burnIn`` / ``inference`` / ``learn``: the reference fans
``gibbsthread`` / ``learnthread`` out over a thread pool (run_pool, factorgraph.py:13-24, called
at :141, :163, :202); here those three call sites go through the C-ABI to HIP kernels on the
MI355X.  The STATE arrays -- ``var_value``, ``var_value_evid``, ``weight_value``, ``count`` -- are
uploaded when a call starts and downloaded when it returns, so in-place edits of them between
calls are honoured exactly like in the reference.  The STRUCTURE (``variable`` incl. isEvidence /
initialValue of evidence, ``weight['isFixed']``, ``factor``, ``fmap``, ``vmap``,
``factor_index``) is compiled into the device layout once, at the first call; after editing it in
place call ``invalidate()`` so that the next call recompiles.
"""

import ctypes as C
import sys
import warnings

import numpy as np

from . import _lib
from .timer import Timer


class FactorGraph(object):
    """A factor graph resident on one MI355X.

    Positional parameters are the reference's (factorgraph.py:30-31).  Keyword-only extras:
    ``device`` (HIP ordinal), ``seed`` (Philox key / MT19937 seed), ``scan`` ("chromatic" or
    "sequential"), ``head_by_vid`` (intended head lookup for IMPLY_MLN-type factors instead of
    the literal inference.py:243 indexing), ``own_range`` ((begin, end) variable ids sampled by
    this handle when the graph is range-partitioned over several GPUs), ``learn_cap`` / ``learn_lag``
    (chromatic learning: cap on visits x stepsize per weight and colour class; the weight update of a
    class overlapped with the next class's sampling, nsk_set_learn_lag).
    """

    def __init__(self, weight, variable, factor, fmap, vmap, factor_index, var_copies,
                 weight_copies, fid, workers, *, device=0, seed=0, scan="chromatic",
                 head_by_vid=False, own_range=None, learn_cap=0.5, global_ids=None, learn_lag=True):
        self.weight, self.variable, self.factor = weight, variable, factor
        self.fmap, self.vmap, self.factor_index = fmap, vmap, factor_index

        nvar = variable.shape[0]
        # tally layout: one slot for a binary variable, `cardinality` slots otherwise (native: the records are packed
        # 27-byte structs, numpy's strided field reads take seconds at 50M variables)
        self.cstart = np.zeros(nvar + 1, np.int64)
        init = np.empty(nvar, np.int64)
        max_card, longest = C.c_int64(), C.c_int64()
        vc, mc = _lib.as_c(variable), _lib.as_c(vmap)
        if vc.dtype.itemsize != 27 or mc.dtype.itemsize != 24:
            raise TypeError("variable / vmap records of %d / %d bytes, expected 27 / 24 (numbskulltypes)" % (vc.dtype.itemsize, mc.dtype.itemsize))
        _lib.check(_lib.lib().nsk_state_layout(nvar, _lib.ptr(vc), len(mc), _lib.ptr(mc), _lib.ptr(self.cstart), _lib.ptr(init),
                                               C.byref(max_card), C.byref(longest)))
        ncount = int(self.cstart[nvar])
        self.count = np.zeros(ncount, np.int64)

        self.var_value_evid = np.tile(init, (var_copies, 1))
        self.var_value = np.tile(init, (var_copies, 1))
        self.weight_value = np.tile(weight["initialValue"], (weight_copies, 1))

        # scratch arrays of the reference's CPU threads; kept for attribute compatibility only
        self.Z = np.zeros((workers, int(max_card.value) if nvar else 0))
        self.fids = np.zeros((workers, 2 * int(longest.value)), factor_index.dtype)

        self.fid = fid
        assert workers > 0
        self.threads = workers          # accepted for compatibility; the GPU ignores it
        self.threadpool = None
        self.marginals = np.zeros(ncount)
        self.inference_epoch_time = 0.0
        self.inference_total_time = 0.0
        self.learning_epoch_time = 0.0
        self.learning_total_time = 0.0

        self.device = int(device)
        self.seed = int(seed)
        self.scan = scan
        self.learn_cap = float(learn_cap)
        self.learn_lag = bool(learn_lag)
        self.head_by_vid = bool(head_by_vid)
        self.own_range = own_range
        # shard-local graph (graphgen.extract_shard): global id of every local variable; the shard's
        # generator streams are tagged with the GLOBAL id of its first owned variable
        self.global_ids = None if global_ids is None else np.ascontiguousarray(global_ids, np.int64)
        self._handle = None
        self._keep = None
        self._clipped_seen = 0

    # ------------------------------------------------------------------ device handle
    def _descriptor(self):
        """nsk_graph_desc over the arrays this object owns (plus the arrays kept alive)."""
        arrays = [_lib.as_c(a) for a in (self.weight, self.variable, self.factor, self.fmap,
                                         self.vmap)]
        fi = _lib.as_c(self.factor_index, np.int64)
        sizes = (9, 27, 34, 16, 24)
        for a, s in zip(arrays, sizes):
            if a.dtype.itemsize != s:
                raise TypeError("record array with itemsize %d, expected %d" % (a.dtype.itemsize, s))
        w, v, f, fm, vm = arrays
        ob, oe = self.own_range if self.own_range is not None else (0, 0)
        desc = _lib.GraphDesc(len(w), len(v), len(f), len(fm), len(vm), len(fi),
                              w.ctypes.data, v.ctypes.data, f.ctypes.data, fm.ctypes.data,
                              vm.ctypes.data, fi.ctypes.data,
                              (_lib.FLAG_HEAD_BY_VID if self.head_by_vid else 0) |
                              (_lib.FLAG_PARTITION if self.own_range is not None else 0), self.device,
                              int(ob), int(oe))
        return desc, (arrays, fi)

    def plan(self):
        """Host-only: validate and colour the graph as the device build would (no GPU needed).
        Returns (color[nvar] with -1 for variables this handle does not sample, info dict)."""
        desc, keep = self._descriptor()
        color = np.zeros(self.variable.shape[0], np.int32)
        inf = _lib.GraphInfo()
        _lib.check(_lib.lib().nsk_graph_plan(C.byref(desc), _lib.ptr(color), C.byref(inf)))
        return color, {k: getattr(inf, k) for k, _ in inf._fields_}

    def ghost_needs(self, host_only=False):
        """Sorted ids of the variables outside ``own_range`` that this partition's variables read
        (what the boundary exchange must deliver).  ``host_only`` plans without touching a GPU."""
        L = _lib.lib()
        n = C.c_int64()
        if host_only:
            desc, keep = self._descriptor()
            _lib.check(L.nsk_graph_plan_needs(C.byref(desc), C.byref(n), None))
            out = np.zeros(n.value, np.int32)
            _lib.check(L.nsk_graph_plan_needs(C.byref(desc), C.byref(n), _lib.ptr(out)))
            return out
        h = self._engine()
        _lib.check(L.nsk_ghost_needs(h, C.byref(n), None))
        out = np.zeros(n.value, np.int32)
        _lib.check(L.nsk_ghost_needs(h, C.byref(n), _lib.ptr(out)))
        return out

    def _engine(self):
        """Create (once) the device-side graph.  Fails loudly without a GPU."""
        if self._handle is not None:
            return self._handle
        L = _lib.lib()
        desc, keep = self._descriptor()
        h = C.c_void_p()
        _lib.check(L.nsk_graph_create(C.byref(desc), C.byref(h)))
        self._handle = h
        self._keep = keep
        _lib.check(L.nsk_set_seed(h, self.seed, 0))
        if self.global_ids is not None and self.own_range is not None and len(self.global_ids):
            lo = int(self.own_range[0])
            tag = int(self.global_ids[lo]) if lo < len(self.global_ids) else int(self.global_ids[-1]) + 1
            _lib.check(L.nsk_set_rng_tag(h, tag & 0xFFFFFFFF))
        scan = {"chromatic": _lib.SCAN_CHROMATIC, "sequential": _lib.SCAN_SEQUENTIAL}[self.scan]
        _lib.check(L.nsk_set_scan(h, scan))
        _lib.check(L.nsk_set_learn_cap(h, self.learn_cap))
        _lib.check(L.nsk_set_learn_lag(h, int(self.learn_lag)))
        return h

    def close(self):
        if self._handle is not None:
            _lib.lib().nsk_graph_destroy(self._handle)
            self._handle = None
            self._clipped_seen = 0

    def invalidate(self):
        """Drop the compiled device graph: the next burnIn / inference / learn call recompiles it
        from the current contents of the structural arrays (the reference reads them on every
        sweep, e.g. after its master marks ownership with isEvidence == 4)."""
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed, sweep0=0):
        self.seed = int(seed)
        if self._handle is not None:
            _lib.check(_lib.lib().nsk_set_seed(self._handle, self.seed, int(sweep0)))

    def info(self):
        inf = _lib.GraphInfo()
        _lib.check(_lib.lib().nsk_graph_get_info(self._engine(), C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def layout(self):
        """Internal id of every variable (its index in the device's value arrays; for a sampled
        variable its position in the compiled layout = its generator id in the chromatic scan)."""
        iid = np.zeros(self.variable.shape[0], np.int32)
        nid = C.c_int64()
        _lib.check(_lib.lib().nsk_graph_get_layout(self._engine(), _lib.ptr(iid), C.byref(nid)))
        return iid.astype(np.int64)

    def generators(self):
        """The chromatic scan's generator of every variable (nsk_graph_get_generators): position in the
        compiled layout | quad-scheme flag << 40; -1 for variables this handle does not sample."""
        out = np.zeros(self.variable.shape[0], np.int64)
        _lib.check(_lib.lib().nsk_graph_get_generators(self._engine(), _lib.ptr(out)))
        return out

    def weight_slots(self):
        """Slot of every weight in the device table (nsk_graph_get_weight_slots); the identity unless the
        handle renumbered its single-factor weights."""
        out = np.zeros(self.weight.shape[0], np.int64)
        rc = _lib.lib().nsk_graph_get_weight_slots(self._engine(), _lib.ptr(out))
        if rc < 0:
            _lib.check(rc)
        return out

    def colors(self):
        out = np.zeros(self.variable.shape[0], np.int32)
        _lib.check(_lib.lib().nsk_graph_get_colors(self._engine(), _lib.ptr(out)))
        return out

    def _push(self, var_copy, weight_copy):
        vv = _lib.as_c(self.var_value[var_copy], np.int64)
        ve = _lib.as_c(self.var_value_evid[var_copy], np.int64)
        wv = _lib.as_c(self.weight_value[weight_copy], np.float64)
        cnt = _lib.as_c(self.count, np.int64)
        _lib.check(_lib.lib().nsk_state_upload(self._engine(), _lib.ptr(vv), _lib.ptr(ve),
                                               _lib.ptr(wv), _lib.ptr(cnt)))

    def _pull(self, var_copy, weight_copy, values=True, weights=True, count=True):
        """Download state into the arrays the caller sees.  Straight into their memory when it is a
        C-contiguous array of the right type (the usual case: rows of ``var_value`` etc.) -- a fresh
        buffer per call costs more in page faults than the transfer itself (240 MB at 10M variables)."""
        def target(a, dtype, n):
            # (the library writes n elements: an array the caller replaced by a shorter one takes the
            # temporary buffer and fails in the assignment below, as it did in the reference)
            ok = (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.flags.writeable
                  and a.shape == (n,))
            return (a, None) if ok else (np.empty(n, dtype), a)
        nv, nw, nc = self.variable.shape[0], self.weight.shape[0], int(self.cstart[-1]) if len(self.cstart) else 0
        vv, vv_to = target(self.var_value[var_copy], np.int64, nv) if values else (None, None)
        ve, ve_to = target(self.var_value_evid[var_copy], np.int64, nv) if values else (None, None)
        wv, wv_to = target(self.weight_value[weight_copy], np.float64, nw) if weights else (None, None)
        cnt, cnt_to = target(self.count, np.int64, nc) if count else (None, None)
        _lib.check(_lib.lib().nsk_state_download(self._engine(), _lib.ptr(vv), _lib.ptr(ve),
                                                 _lib.ptr(wv), _lib.ptr(cnt)))
        for buf, to in ((vv, vv_to), (ve, ve_to), (wv, wv_to), (cnt, cnt_to)):
            if to is not None:
                to[:] = buf

    # ------------------------------------------------------------------ reference API
    def clear(self):
        self.count[:] = 0

    def getWeights(self, weight_copy=0):
        return self.weight_value[weight_copy][:]

    def getMarginals(self, varIds=None):
        return self.marginals if not varIds else self.marginals[varIds]

    def diagnostics(self, epochs):
        print('Inference took %.03f sec.' % self.inference_total_time)
        epochs = epochs or 1
        bins = 10
        assert self.count.min(initial=0) >= 0
        assert self.count.max(initial=0) <= epochs
        which = np.minimum(self.count * bins // epochs, bins - 1)
        hist = np.bincount(which, minlength=bins)
        for i in range(bins):
            print("Prob. " + str(i / 10.0) + ".." + str((i + 1) / 10.0) + ": \
                  " + str(hist[i]) + " variables")

    def diagnosticsLearning(self, weight_copy=0):
        print('Learning epoch took %.03f sec.' % self.learning_epoch_time)
        print("Weights:")
        for i, w in enumerate(self.weight):
            print("    weightId:", i)
            print("        isFixed:", w["isFixed"])
            print("        weight: ", self.weight_value[weight_copy][i])
            print()

    def _sweep(self, nsweeps, sample_evidence, burnin):
        h = self._engine()
        _lib.check(_lib.lib().nsk_gibbs_sweeps(h, int(nsweeps), int(bool(sample_evidence)),
                                               int(bool(burnin))))

    def burnIn(self, epochs, sample_evidence, diagnostics=False, var_copy=0, weight_copy=0):
        """factorgraph.py:129-143 -- `epochs` sweeps that do not touch the tally."""
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": STARTED BURN-IN...")
        if epochs > 0:
            self._push(var_copy, weight_copy)
            self._sweep(epochs, sample_evidence, True)
            self._pull(var_copy, weight_copy, weights=False, count=False)
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": DONE WITH BURN-IN")

    def inference(self, burnin_epochs, epochs, sample_evidence=False, diagnostics=False,
                  var_copy=0, weight_copy=0):
        """factorgraph.py:145-175."""
        if burnin_epochs > 0:
            self.burnIn(burnin_epochs, sample_evidence, diagnostics=diagnostics,
                        var_copy=var_copy, weight_copy=weight_copy)
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": STARTED INFERENCE")
        if epochs > 0:
            L, h = _lib.lib(), self._engine()
            self._push(var_copy, weight_copy)
            if diagnostics:     # per-epoch timing lines, like the reference prints them
                for ep in range(epochs):
                    with Timer() as timer:
                        self._sweep(1, sample_evidence, False)
                        _lib.check(L.nsk_synchronize(h))
                    self.inference_epoch_time = timer.interval
                    self.inference_total_time += timer.interval
                    print('Inference epoch #%d took %.03f sec.' % (ep, self.inference_epoch_time))
            else:
                with Timer() as timer:
                    self._sweep(epochs, sample_evidence, False)
                    _lib.check(L.nsk_synchronize(h))
                self.inference_epoch_time = timer.interval / epochs
                self.inference_total_time += timer.interval
            self._pull(var_copy, weight_copy, weights=False)
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": DONE WITH INFERENCE")
        if epochs != 0:
            self.marginals = self.count / float(epochs)
        if diagnostics:
            self.diagnostics(epochs)

    def learn(self, burnin_epochs, epochs, stepsize, decay, regularization, reg_param, truncation,
              diagnostics=False, verbose=False, learn_non_evidence=False, var_copy=0,
              weight_copy=0):
        """factorgraph.py:177-208."""
        if burnin_epochs > 0:
            self.burnIn(burnin_epochs, True, diagnostics=diagnostics, var_copy=var_copy,
                        weight_copy=weight_copy)
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": STARTED LEARNING")
        if epochs > 0:
            L, h = _lib.lib(), self._engine()
            self._push(var_copy, weight_copy)
            args = (int(regularization), float(reg_param), int(truncation),
                    int(bool(learn_non_evidence)))
            if diagnostics:
                for ep in range(epochs):
                    print("FACTOR " + str(self.fid) + ": EPOCH #" + str(ep))
                    print("Current stepsize = " + str(stepsize))
                    if verbose:
                        self._pull(var_copy, weight_copy, values=False, count=False)
                        self.diagnosticsLearning(weight_copy)
                    sys.stdout.flush()
                    with Timer() as timer:
                        _lib.check(L.nsk_learn_sweeps(h, 1, float(stepsize), float(decay), *args))
                        _lib.check(L.nsk_synchronize(h))
                    self.learning_epoch_time = timer.interval
                    self.learning_total_time += timer.interval
                    stepsize *= decay
            else:
                with Timer() as timer:
                    _lib.check(L.nsk_learn_sweeps(h, int(epochs), float(stepsize), float(decay),
                                                  *args))
                    _lib.check(L.nsk_synchronize(h))
                self.learning_epoch_time = timer.interval / epochs
                self.learning_total_time += timer.interval
            self._pull(var_copy, weight_copy, count=False)
            # Chromatic learning applies the SGD rule once per colour class and caps visits * step
            # of a weight at `learn_cap` (a weight tied to many factors then moves more slowly per
            # epoch than under the reference's per-visit rule, learning.py:110-125; same fixed
            # point).  Say so instead of leaving it to a counter nobody reads.
            clipped = self.info()["learn_clipped"]
            if self.scan == "chromatic" and clipped > self._clipped_seen:
                msg = ("numbskull_amd: %d weight update(s) used a step below stepsize because one "
                       "colour class visits the weight more than learn_cap / stepsize = %g times "
                       "(learn_cap=%g; pass learn_cap=0 for the uncapped batch rule, or "
                       "scan='sequential' for the reference's per-visit trajectory)"
                       % (clipped - self._clipped_seen, self.learn_cap / max(stepsize, 1e-300),
                          self.learn_cap))
                warnings.warn(msg, RuntimeWarning, stacklevel=2)
                if diagnostics:
                    print(msg)
            self._clipped_seen = clipped
        if diagnostics:
            print("FACTOR " + str(self.fid) + ": DONE WITH LEARNING")

    def dump_weights(self, fout, weight_copy=0):
        """<wid, weight> text file (factorgraph.py:210-214)."""
        w = self.weight_value[weight_copy]
        with open(fout, 'w') as out:
            out.write("".join('%d %f\n' % (i, w[i]) for i in range(self.weight.shape[0])))

    def dump_probabilities(self, fout, epochs):
        """<vid, value, prob> text file (factorgraph.py:216-229): binary variables print the
        probability of value 1, others one line per domain value (written natively,
        nsk_write_probabilities)."""
        epochs = epochs or 1
        v, vm = _lib.as_c(self.variable), _lib.as_c(self.vmap)
        cs, cnt = _lib.as_c(self.cstart, np.int64), _lib.as_c(self.count, np.int64)
        _lib.check(_lib.lib().nsk_write_probabilities(str(fout).encode(), len(v), _lib.ptr(v), _lib.ptr(vm),
                                                      len(vm), _lib.ptr(cs), _lib.ptr(cnt), len(cnt),
                                                      float(epochs)))

