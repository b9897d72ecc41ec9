"""ctypes binding of the CPU oracle (oracle/libnsk_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
nothing under numbskull_amd/ does (tests/test_layout.py enforces it).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnsk_oracle.so")

OK, E_FACTOR_FUNC, E_INDEX = 0, -2, -3


class _Graph(C.Structure):
    _fields_ = [("nvar", C.c_int64), ("nfactor", C.c_int64), ("nweight", C.c_int64),
                ("nedge", C.c_int64), ("nvtf", C.c_int64),
                ("weight", C.c_void_p), ("variable", C.c_void_p), ("factor", C.c_void_p),
                ("fmap", C.c_void_p), ("vmap", C.c_void_p), ("factor_index", C.c_void_p),
                ("head_by_vid", C.c_int), ("rng_id", C.c_void_p), ("grad_shift", C.c_int),
                ("rng_tag", C.c_uint32)]


class _MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile) if the shared object is missing/stale."""
    src = os.path.join(_HERE, "nsk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libnsk_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_override = None


def use_library(path):
    """Load the oracle from another shared object (bench.py builds a -march=native copy on the
    node it times the CPU baseline on)."""
    global _lib, _override
    _override, _lib = path, None


def lib():
    global _lib
    if _lib is None:
        if _override is None:
            build()
        _lib = C.CDLL(_override or _SO)
        _lib.orc_exp_det.restype = C.c_double
        _lib.orc_exp_det.argtypes = [C.c_double]
        _lib.orc_mt_res53.restype = C.c_double
        _lib.orc_u53.restype = C.c_double
        _lib.orc_u53.argtypes = [C.c_uint32, C.c_uint32]
        # the device-mode sweeps walk a colour class with the host's threads (same results whatever their number:
        # tests/test_oracle_goldens.py::test_device_mode_is_thread_count_independent)
        _lib.orc_set_threads(C.c_int(min(64, os.cpu_count() or 1)))
    return _lib


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


class MT:
    """MT19937 stream seeded like np.random.seed(s) (kind='numpy') or random.seed(s)."""

    def __init__(self, seed, kind="numpy"):
        self.s = _MT()
        if kind == "numpy":
            lib().orc_mt_seed_numpy(C.byref(self.s), C.c_uint32(seed))
        else:
            lib().orc_mt_seed_python(C.byref(self.s), C.c_uint64(seed))

    def random(self):
        return lib().orc_mt_res53(C.byref(self.s))


def exp_det(x):
    x = np.ascontiguousarray(x, np.float64)
    f = lib().orc_exp_det
    return np.array([f(float(v)) for v in x.ravel()]).reshape(x.shape)


def philox(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1),
                         C.c_uint32(c2), C.c_uint32(c3), out)
    return [int(v) for v in out]


class Graph:
    """Borrowed view of the reference-layout record arrays (kept alive by this object)."""

    def __init__(self, weight, variable, factor, fmap, vmap, factor_index, head_by_vid=False):
        self.weight = np.ascontiguousarray(weight)
        self.variable = np.ascontiguousarray(variable)
        self.factor = np.ascontiguousarray(factor)
        self.fmap = np.ascontiguousarray(fmap)
        self.vmap = np.ascontiguousarray(vmap)
        self.factor_index = np.ascontiguousarray(factor_index, np.int64)
        assert self.weight.dtype.itemsize == 9 and self.variable.dtype.itemsize == 27
        assert self.factor.dtype.itemsize == 34 and self.fmap.dtype.itemsize == 16
        assert self.vmap.dtype.itemsize == 24
        self.g = _Graph(len(self.variable), len(self.factor), len(self.weight), len(self.fmap),
                        len(self.vmap), self.weight.ctypes.data, self.variable.ctypes.data,
                        self.factor.ctypes.data, self.fmap.ctypes.data, self.vmap.ctypes.data,
                        self.factor_index.ctypes.data, int(bool(head_by_vid)), None, 0, 0)
        self.rng_id = None
        self.device_lag = None          # nsk_graph_info.learn_lag of the handle this oracle shadows (tests/util.oracle_of)
        card = self.variable["cardinality"]
        self.cstart = np.zeros(len(card) + 1, np.int64)
        self.cstart[1:] = np.where(card == 2, 1, card)
        np.cumsum(self.cstart, out=self.cstart)
        self.maxcard = int(card.max()) if len(card) else 1
        self.maxlist = int(self.vmap["factor_index_length"].max()) if len(self.vmap) else 0

    def set_rng_ids(self, ids):
        """Device mode: the generator id of every variable = its position in the library's layout
        (FactorGraph.layout()); None = the variable id."""
        if ids is None:
            self.rng_id, self.g.rng_id = None, None
        else:
            self.rng_id = np.ascontiguousarray(ids, np.int64)
            assert len(self.rng_id) == len(self.variable)
            self.g.rng_id = self.rng_id.ctypes.data

    def set_grad_shift(self, s):
        """Device-mode learning: fraction bits traded for range in the fixed-point gradient sums
        (nsk_graph_info.grad_shift)."""
        self.g.grad_shift = int(s)

    def set_rng_tag(self, tag):
        """Device mode: the shard tag of the emulated handle (its first owned variable id; 0 for a
        handle that owns the whole graph) -- XORed into Philox counter word 3."""
        self.g.rng_tag = int(tag) & 0xFFFFFFFF

    # ---- state helpers (factorgraph.py:41-53) ----
    def initial_state(self):
        vv = self.variable["initialValue"].astype(np.int64).copy()
        return vv, vv.copy(), self.weight["initialValue"].astype(np.float64).copy(), \
            np.zeros(int(self.cstart[-1]), np.int64)

    def eval_factor(self, fid, var_samp, value, var_value):
        out = C.c_double()
        rc = lib().orc_eval_factor(C.byref(self.g), C.c_int64(fid), C.c_int64(var_samp),
                                   C.c_int64(value), _p(var_value), C.byref(out))
        return rc, out.value

    def potential(self, var_samp, value, var_value, weight_value):
        out = C.c_double()
        rc = lib().orc_potential(C.byref(self.g), C.c_int64(var_samp), C.c_int64(value),
                                 _p(var_value), _p(weight_value), C.byref(out))
        return rc, out.value

    # ---- reference mode ----
    def gibbs_ref(self, np_rng, var_value, weight_value, count, sample_evidence=True,
                  burnin=False, shard=0, nshards=1):
        Z = np.zeros(self.maxcard)
        return lib().orc_gibbs_shard_ref(
            C.byref(self.g), C.c_int64(shard), C.c_int64(nshards), _p(Z), _p(self.cstart),
            _p(count), _p(var_value), _p(weight_value), int(sample_evidence), int(burnin),
            C.byref(np_rng.s))

    def learn_ref(self, np_rng, py_rng, var_value, var_value_evid, weight_value, step,
                  regularization, reg_param, truncation, learn_non_evidence, shard=0, nshards=1):
        Z = np.zeros(self.maxcard)
        fids = np.zeros(2 * self.maxlist + 1, np.int64)
        return lib().orc_learn_shard_ref(
            C.byref(self.g), C.c_int64(shard), C.c_int64(nshards), C.c_double(step),
            int(regularization), C.c_double(reg_param), C.c_int64(truncation), _p(Z), _p(fids),
            _p(var_value), _p(var_value_evid), _p(weight_value), int(learn_non_evidence),
            C.byref(np_rng.s), C.byref(py_rng.s))

    # ---- device mode ----
    def gibbs_dev(self, order, phase_start, var_value, weight_value, count, seed, sweep,
                  sample_evidence=True, burnin=False):
        order = np.ascontiguousarray(order, np.int64)
        phase_start = np.ascontiguousarray(phase_start, np.int64)
        return lib().orc_gibbs_sweep_dev(
            C.byref(self.g), _p(order), _p(phase_start), C.c_int64(len(phase_start) - 1),
            _p(self.cstart), _p(count), _p(var_value), _p(weight_value), int(sample_evidence),
            int(burnin), C.c_uint64(seed), C.c_uint64(sweep))

    def learn_dev(self, order, phase_start, var_value, var_value_evid, weight_value, step,
                  regularization, reg_param, truncation, learn_non_evidence, seed, sweep, cap=0.5, lag=None):
        """One learning sweep in device mode.  ``lag``: the device's one-class lag (nsk_set_learn_lag,
        its default) -- a float64 array the caller sets to ``weight_value.copy()`` at the start of every
        device call (a call drains the pipeline) and hands to every sweep of that call; None = every
        class sees the previous class's update."""
        assert lag is None or (lag.dtype == np.float64 and lag.flags.c_contiguous and len(lag) == len(weight_value))
        order = np.ascontiguousarray(order, np.int64)
        phase_start = np.ascontiguousarray(phase_start, np.int64)
        return lib().orc_learn_sweep_dev(
            C.byref(self.g), _p(order), _p(phase_start), C.c_int64(len(phase_start) - 1),
            C.c_double(step), int(regularization), C.c_double(reg_param), C.c_int64(truncation),
            _p(var_value), _p(var_value_evid), _p(weight_value), int(learn_non_evidence),
            C.c_uint64(seed), C.c_uint64(sweep), C.c_double(cap), _p(lag))

    def learn_call(self, order, phase_start, var_value, var_value_evid, weight_value, nsweeps, step, decay,
                   regularization, reg_param, truncation, learn_non_evidence, seed, sweep0, cap=0.5, lag=None):
        """What ONE nsk_learn_sweeps call does: ``nsweeps`` epochs from sweep index ``sweep0``, step *= decay
        after each (factorgraph.py:206), the one-class lag pipeline (``lag``) started from the call's weights
        and drained into them at its end.  ``lag=None``: the device's default -- lagged iff the graph has at
        most 256 weights (nsk_graph_info.learn_lag; ``self.device_lag`` when oracle_of set it)."""
        if lag is None:
            lag = self.device_lag if self.device_lag is not None else len(weight_value) <= 256
        lagw = np.ascontiguousarray(weight_value, np.float64).copy() if lag else None
        for s in range(nsweeps):
            rc = self.learn_dev(order, phase_start, var_value, var_value_evid, weight_value, step, regularization,
                                reg_param, truncation, learn_non_evidence, seed, sweep0 + s, cap, lagw)
            if rc:
                return rc
            step *= decay
        return 0

    def check_coloring(self, color):
        """(-1, -1) when no sampled variable reads a variable of its own colour, else such a pair."""
        color = np.ascontiguousarray(color, np.int32)
        assert len(color) == len(self.variable)
        other = C.c_int64(-1)
        f = lib().orc_check_coloring
        f.restype = C.c_int64
        v = f(C.byref(self.g), _p(color), C.byref(other))
        return int(v), int(other.value)

    # ---- CPU baseline (Hogwild threads) ----
    def gibbs_hogwild(self, nthreads, nsweeps, var_value, weight_value, count, seed,
                      sample_evidence=True, burnin=False):
        return lib().orc_gibbs_hogwild(
            C.byref(self.g), int(nthreads), C.c_int64(nsweeps), _p(self.cstart), _p(count),
            _p(var_value), _p(weight_value), int(sample_evidence), int(burnin), C.c_uint32(seed))

    def learn_hogwild(self, nthreads, nsweeps, var_value, var_value_evid, weight_value, step,
                      decay, regularization, reg_param, truncation, learn_non_evidence, seed):
        return lib().orc_learn_hogwild(
            C.byref(self.g), int(nthreads), C.c_int64(nsweeps), C.c_double(step),
            C.c_double(decay), int(regularization), C.c_double(reg_param), C.c_int64(truncation),
            _p(var_value), _p(var_value_evid), _p(weight_value), int(learn_non_evidence),
            C.c_uint32(seed))


def compute_var_map(variable, factor, fmap, domain_mask, factors_to_skip=None, vmap_values=None):
    """loadFactorGraph's index build (numbskull.py:217-238 + dataloading.py:16-81).
    Returns (variable_with_vtf_offset, vmap, factor_index, rc)."""
    from numpy import dtype
    VTF = dtype([("value", np.int64), ("factor_index_offset", np.int64),
                 ("factor_index_length", np.int64)])
    variable = variable.copy()
    skip = np.ascontiguousarray(factors_to_skip if factors_to_skip is not None else [], np.int64)
    per = np.where(variable["dataType"] == 0, 1, variable["cardinality"]).astype(np.int64)
    variable["vtf_offset"] = np.cumsum(per) - per
    nvtf = int(per.sum())
    vmap = np.zeros(nvtf, VTF)
    if vmap_values is not None:
        vmap["value"] = vmap_values
    nfi = int(factor["arity"].sum() - factor["arity"][skip].sum())
    factor_index = np.zeros(nfi, np.int64)
    dm = np.ascontiguousarray(domain_mask, np.uint8)
    rc = lib().orc_compute_var_map(
        C.c_int64(len(variable)), _p(variable), C.c_int64(len(factor)), _p(factor),
        C.c_int64(len(fmap)), _p(fmap), C.c_int64(nvtf), _p(vmap), C.c_int64(nfi),
        _p(factor_index), _p(dm), _p(skip), C.c_int64(len(skip)))
    return variable, vmap, factor_index, rc
