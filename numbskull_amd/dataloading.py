"""Readers for the DeepDive binary graph format and the inverted-index build.

Function names and argument meaning follow numbskull/dataloading.py (load_weights 103-123,
load_variables 126-156, load_domains 159-187, load_factors 190-237, compute_var_map 16-81), but
nothing is parsed byte-at-a-time: the fixed-width files are decoded with big-endian numpy views,
the variable-length ``graph.factors`` and the index build run natively in the C-ABI library
(nsk_parse_factors / nsk_compute_var_map, host code, no GPU needed).
"""

import numpy as np

from . import _lib
from .numbskulltypes import VarToFactor

_W_DISK = np.dtype([("weightId", ">i8"), ("isFixed", "u1"), ("initialValue", ">f8")])
_V_DISK = np.dtype([("variableId", ">i8"), ("isEvidence", "u1"), ("initialValue", ">i8"),
                    ("dataType", ">i2"), ("cardinality", ">i8")])


def dataType(i):
    return {0: "Boolean", 1: "Categorical"}.get(i, "Unknown")


def assign_vtf_offsets(variable):
    """numbskull.py:221-227 / 311-317: one VarToFactor record for a boolean (dataType 0)
    variable, ``cardinality`` records otherwise.  Returns the number of records."""
    per = np.where(variable["dataType"] == 0, 1, variable["cardinality"]).astype(np.int64)
    variable["vtf_offset"] = np.cumsum(per) - per
    return int(per.sum())


def compute_var_map(variables, factors, fmap, vmap, factor_index, domain_mask,
                    factors_to_skip=np.empty(0, np.int64)):
    """In-place fill of ``vmap`` / ``factor_index`` exactly like dataloading.py:16-81."""
    v = _lib.as_c(variables)
    f = _lib.as_c(factors)
    fm = _lib.as_c(fmap)
    dm = _lib.as_c(domain_mask, np.uint8)
    skip = _lib.as_c(factors_to_skip, np.int64)
    if not (vmap.flags.c_contiguous and factor_index.flags.c_contiguous):
        raise ValueError("vmap and factor_index must be C-contiguous (they are filled in place)")
    _lib.check(_lib.lib().nsk_compute_var_map(
        len(v), _lib.ptr(v), len(f), _lib.ptr(f), len(fm), _lib.ptr(fm), len(vmap), _lib.ptr(vmap),
        len(factor_index), _lib.ptr(factor_index), _lib.ptr(dm), _lib.ptr(skip), len(skip)))


def load_weights(data, nweights, weights):
    """graph.weights: 17-byte records placed by their weightId (dataloading.py:103-123)."""
    rec = np.frombuffer(data, _W_DISK, count=nweights)
    ids = rec["weightId"].astype(np.int64)
    weights["isFixed"][ids] = rec["isFixed"] != 0
    weights["initialValue"][ids] = rec["initialValue"]
    print("LOADED WEIGHTS")


def load_variables(data, nvariables, variables):
    """graph.variables: 27-byte records placed by their variableId (dataloading.py:126-156)."""
    rec = np.frombuffer(data, _V_DISK, count=nvariables)
    ids = rec["variableId"].astype(np.int64)
    variables["isEvidence"][ids] = rec["isEvidence"].astype(np.int8)
    variables["initialValue"][ids] = rec["initialValue"]
    variables["dataType"][ids] = rec["dataType"]
    variables["cardinality"][ids] = rec["cardinality"]
    print("LOADED VARS")


def load_domains(data, domain_mask, vmap, variables):
    """graph.domains: (vid, cardinality, values...) blocks; marks the variable, stores its sorted
    domain in ``vmap.value`` and rewrites its initialValue as a dense index
    (dataloading.py:159-187; parsed natively, nsk_parse_domains)."""
    raw = np.frombuffer(data, np.uint8)
    dm = domain_mask.view(np.uint8) if domain_mask.dtype == np.bool_ else domain_mask
    if not (dm.flags.c_contiguous and variables.flags.c_contiguous and vmap.flags.c_contiguous):
        raise ValueError("load_domains needs contiguous arrays (they are filled in place)")
    _lib.check(_lib.lib().nsk_parse_domains(_lib.ptr(raw), len(raw), _lib.ptr(dm), len(variables),
                                            _lib.ptr(variables), _lib.ptr(vmap), len(vmap)))
    print("LOADED DOMAINS")


def load_factors(data, nfactors, factors, fmap, domain_mask, variable, vmap):
    """graph.factors: variable-length records, parsed natively (dataloading.py:190-237)."""
    raw = np.frombuffer(data, np.uint8)
    dm = _lib.as_c(domain_mask, np.uint8)
    v = _lib.as_c(variable)
    vm = _lib.as_c(vmap)
    _lib.check(_lib.lib().nsk_parse_factors(_lib.ptr(raw), len(raw), nfactors, len(fmap),
                                            _lib.ptr(factors), _lib.ptr(fmap), _lib.ptr(dm),
                                            _lib.ptr(v), len(v), _lib.ptr(vm)))
    print("LOADED FACTORS")


def new_index(variable, nedges):
    """Allocate (vmap, factor_index) for a graph whose vtf offsets are assigned."""
    nvtf = assign_vtf_offsets(variable)
    return np.zeros(nvtf, VarToFactor), np.zeros(int(nedges), np.int64)
