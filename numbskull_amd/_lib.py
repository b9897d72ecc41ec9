"""ctypes binding of libnumbskull_amd.so (the C-ABI declared in include/numbskull_amd.h).

There is no CPU fallback: if the shared library is missing the import of any compute path fails
loudly with build instructions, and if no MI355X is visible ``nsk_graph_create`` returns
NSK_E_DEVICE, raised here as RuntimeError.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSK_LIB") or os.path.join(_HERE, "libnumbskull_amd.so")   # NSK_LIB: ablation builds (tools/)

OK, E_INVALID, E_FACTOR_FUNC, E_INDEX, E_DEVICE, E_RANGE, E_NOMEM = 0, -1, -2, -3, -4, -5, -6
FLAG_HEAD_BY_VID = 1
FLAG_PARTITION = 2
SCAN_CHROMATIC, SCAN_SEQUENTIAL = 0, 1
BUF_VALUE, BUF_VALUE_EVID, BUF_WEIGHT, BUF_SEND, BUF_RECV, BUF_SEND_EVID, BUF_RECV_EVID = range(7)

# every symbol include/numbskull_amd.h declares (tests/test_cabi.py checks the export list)
SYMBOLS = (
    "nsk_graph_create", "nsk_graph_destroy", "nsk_state_upload", "nsk_state_download",
    "nsk_set_seed", "nsk_set_rng_tag", "nsk_set_scan", "nsk_set_learn_cap", "nsk_set_learn_lag", "nsk_gibbs_sweeps", "nsk_learn_sweeps", "nsk_graph_get_info",
    "nsk_graph_get_colors", "nsk_graph_get_layout", "nsk_graph_get_generators", "nsk_graph_get_weight_slots", "nsk_graph_plan", "nsk_graph_plan_needs", "nsk_profile_begin", "nsk_profile_end", "nsk_device_buffer",
    "nsk_set_stream", "nsk_synchronize", "nsk_ghost_needs", "nsk_exchange_setup", "nsk_exchange_pack",
    "nsk_exchange_unpack", "nsk_comm_unique_id", "nsk_comm_init", "nsk_gibbs_sweeps_exchange",
    "nsk_learn_sweeps_exchange", "nsk_pf_setup", "nsk_p2p_setup", "nsk_p2p_export", "nsk_p2p_import", "nsk_p2p_import_local",
    "nsk_gibbs_sweeps_p2p", "nsk_learn_sweeps_p2p", "nsk_p2p_exchange", "nsk_p2p_selftest", "nsk_p2p_fuse", "nsk_p2p_check", "nsk_p2p_reset", "nsk_profile_mark", "nsk_profile_read", "nsk_graph_order", "nsk_comm_volume", "nsk_graph_partition", "nsk_compute_var_map", "nsk_state_layout", "nsk_parse_factors", "nsk_parse_domains", "nsk_write_probabilities",
    "nsk_selftest_exp", "nsk_selftest_philox", "nsk_selftest_stream", "nsk_device_count", "nsk_last_error", "nsk_version",
)


class GraphDesc(C.Structure):
    _fields_ = [("nweight", C.c_int64), ("nvar", C.c_int64), ("nfactor", C.c_int64),
                ("nedge", C.c_int64), ("nvtf", C.c_int64), ("nfactor_index", C.c_int64),
                ("weight", C.c_void_p), ("variable", C.c_void_p), ("factor", C.c_void_p),
                ("fmap", C.c_void_p), ("vmap", C.c_void_p), ("factor_index", C.c_void_p),
                ("flags", C.c_int32), ("device", C.c_int32),
                ("own_begin", C.c_int64), ("own_end", C.c_int64)]


class GraphInfo(C.Structure):
    _fields_ = [("nvar", C.c_int64), ("nowned", C.c_int64), ("ncolors", C.c_int64),
                ("value_bytes", C.c_int64), ("device_bytes", C.c_int64),
                ("nfast", C.c_int64), ("ngeneric", C.c_int64),
                ("alg_bytes_inference", C.c_double), ("alg_bytes_learning", C.c_double),
                ("sweeps_done", C.c_int64), ("layout_bytes_inference", C.c_double),
                ("layout_bytes_learning", C.c_double), ("ztab_entries", C.c_int64),
                ("compile_seconds", C.c_double), ("learn_cap", C.c_double),
                ("learn_clipped", C.c_int64), ("grad_shift", C.c_int64),
                ("acc_copies", C.c_int64), ("learn_lag", C.c_int64), ("direct_weights", C.c_int64),
                ("weight_slots", C.c_int64), ("layout_hash", C.c_int64), ("p2p_fused", C.c_int64),
                ("tab_quads", C.c_int64), ("wide_quads", C.c_int64)]


_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "numbskull_amd: %s is missing. Build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or "
                "make -C numbskull_amd/csrc); there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.nsk_last_error.restype = C.c_char_p
        L.nsk_version.restype = C.c_char_p
        for name in SYMBOLS:
            if not hasattr(L, name):
                if not os.environ.get("NSK_LIB"):      # (an older ablation build under NSK_LIB may lack newer entry points)
                    raise AttributeError("libnumbskull_amd.so lacks %s: header and library disagree" % name)
                # a call of a missing entry point FAILS (it used to report success: an A/B run against
                # an older build then compared different semantics without saying so)
                def missing(*a, _name=name):
                    raise AttributeError("%s lacks %s (an older build under NSK_LIB)" % (LIB_PATH, _name))
                setattr(L, name, missing)
        L.nsk_gibbs_sweeps.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.nsk_learn_sweeps.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int,
                                       C.c_double, C.c_int64, C.c_int]
        L.nsk_set_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.nsk_set_rng_tag.argtypes = [C.c_void_p, C.c_uint32]
        L.nsk_set_scan.argtypes = [C.c_void_p, C.c_int]
        L.nsk_set_learn_cap.argtypes = [C.c_void_p, C.c_double]
        L.nsk_set_learn_lag.argtypes = [C.c_void_p, C.c_int]
        L.nsk_state_upload.argtypes = [C.c_void_p] * 5
        L.nsk_state_download.argtypes = [C.c_void_p] * 5
        L.nsk_graph_create.argtypes = [C.POINTER(GraphDesc), C.POINTER(C.c_void_p)]
        L.nsk_graph_destroy.argtypes = [C.c_void_p]
        L.nsk_graph_get_info.argtypes = [C.c_void_p, C.POINTER(GraphInfo)]
        L.nsk_graph_get_colors.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_graph_get_layout.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.nsk_graph_get_generators.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_graph_get_weight_slots.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_graph_plan.argtypes = [C.POINTER(GraphDesc), C.c_void_p, C.POINTER(GraphInfo)]
        L.nsk_graph_plan_needs.argtypes = [C.POINTER(GraphDesc), C.POINTER(C.c_int64), C.c_void_p]
        L.nsk_profile_begin.argtypes = [C.c_void_p]
        L.nsk_profile_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.nsk_device_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_int64)]
        L.nsk_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_synchronize.argtypes = [C.c_void_p]
        L.nsk_ghost_needs.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
        L.nsk_exchange_setup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_int64]
        L.nsk_exchange_pack.argtypes = [C.c_void_p, C.c_int]
        L.nsk_exchange_unpack.argtypes = [C.c_void_p, C.c_int]
        L.nsk_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p]
        L.nsk_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p]
        L.nsk_gibbs_sweeps_exchange.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.nsk_learn_sweeps_exchange.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int,
                                                C.c_double, C.c_int64, C.c_int]
        L.nsk_pf_setup.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nsk_p2p_setup.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.nsk_p2p_export.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.nsk_p2p_import.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_p2p_import_local.argtypes = [C.c_void_p, C.c_void_p]
        L.nsk_gibbs_sweeps_p2p.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.nsk_learn_sweeps_p2p.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int,
                                           C.c_double, C.c_int64, C.c_int]
        L.nsk_p2p_exchange.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.nsk_p2p_selftest.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.nsk_p2p_fuse.argtypes = [C.c_void_p, C.c_int]
        L.nsk_p2p_check.argtypes = [C.c_void_p]
        L.nsk_p2p_reset.argtypes = [C.c_void_p]
        L.nsk_profile_mark.argtypes = [C.c_void_p]
        L.nsk_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.nsk_graph_order.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_int64)]
        L.nsk_comm_volume.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int64)]
        L.nsk_graph_partition.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_uint64,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.nsk_compute_var_map.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int64]
        L.nsk_state_layout.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.nsk_parse_factors.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.nsk_parse_domains.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_int64]
        L.nsk_write_probabilities.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_void_p, C.c_int64, C.c_double]
        L.nsk_device_count.argtypes = [C.POINTER(C.c_int)]
        L.nsk_selftest_exp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.nsk_selftest_stream.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.nsk_selftest_philox.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64,
                                          C.c_void_p]
        _lib = L
    return _lib


_EXC = {E_FACTOR_FUNC: NotImplementedError, E_INDEX: IndexError, E_RANGE: OverflowError,
        E_NOMEM: MemoryError, E_INVALID: ValueError, E_DEVICE: RuntimeError}


def check(rc):
    """Map a C-ABI status to the exception the reference would raise."""
    if rc == OK:
        return
    msg = lib().nsk_last_error().decode("utf-8", "replace")
    raise _EXC.get(rc, RuntimeError)(msg or "numbskull_amd error %d" % rc)


def ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def device_count():
    n = C.c_int(0)
    rc = lib().nsk_device_count(C.byref(n))
    return n.value if rc == OK else 0


def as_c(a, dtype=None):
    """C-contiguous view/copy of ``a`` (record arrays keep their packed dtype)."""
    return np.ascontiguousarray(a, dtype)
