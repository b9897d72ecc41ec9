"""Record dtypes of the host-side API.

These are the API types of the reference (numbskull/numbskulltypes.py:11-39): callers build
numpy arrays of these packed records and hand them to ``NumbSkull.loadFactorGraph``.  The
same packed layouts are what the C-ABI (include/numbskull_amd.h: nsk_weight, nsk_variable,
nsk_factor, nsk_ftv, nsk_vtf) reads, so a record array is passed to the library by pointer
without conversion.  The device never sees these AoS records; nsk_graph_create compiles them
into the SoA layout described in DESIGN.md.
"""

import numpy as np

_i8, _f8 = np.int64, np.float64


def _record(*fields):
    return np.dtype(list(fields))


# graph.meta: weights,variables,factors,edges
Meta = _record(("weights", _i8), ("variables", _i8), ("factors", _i8), ("edges", _i8))

# 9 B
Weight = _record(("isFixed", np.bool_), ("initialValue", _f8))

# 27 B; isEvidence: 0 query, 1 evidence, 4 "not owned by this partition"
Variable = _record(("isEvidence", np.int8), ("initialValue", _i8), ("dataType", np.int16),
                   ("cardinality", _i8), ("vtf_offset", _i8))

# 34 B
Factor = _record(("factorFunction", np.int16), ("weightId", _i8), ("featureValue", _f8),
                 ("arity", _i8), ("ftv_offset", _i8))

# 16 B
FactorToVar = _record(("vid", _i8), ("dense_equal_to", _i8))

# 24 B
VarToFactor = _record(("value", _i8), ("factor_index_offset", _i8), ("factor_index_length", _i8))

# 16 B
UnaryFactorOpt = _record(("vid", _i8), ("weightId", _i8))

assert (Weight.itemsize, Variable.itemsize, Factor.itemsize, FactorToVar.itemsize,
        VarToFactor.itemsize) == (9, 27, 34, 16, 24)
