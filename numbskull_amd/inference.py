"""Factor-function table of the sampler.

Mirrors the public names of the reference module (numbskull/inference.py:74-146): the
``FACTORS`` dict and one ``FUNC_<NAME>`` constant per entry -- callers such as the
reference's loadfg.py iterate ``numbskull.inference.FACTORS``.  The sweep itself
(gibbsthread / draw_sample / potential / eval_factor, inference.py:10-71,149-413) is NOT
implemented here: it runs as HIP kernels behind the C-ABI (numbskull_amd/csrc/, reached
through numbskull_amd.factorgraph.FactorGraph).
"""

# name -> id; semantics per id are tabulated in DESIGN.md ("eval_factor") and implemented
# in csrc/nsk_device.h
_BOOLEAN = (("NOOP", -1), ("IMPLY_NATURAL", 0), ("OR", 1), ("AND", 2), ("EQUAL", 3),
            ("ISTRUE", 4), ("LINEAR", 7), ("RATIO", 8), ("LOGICAL", 9), ("IMPLY_MLN", 13))
_CATEGORICAL = (("AND_CAT", 12), ("OR_CAT", 14), ("EQUAL_CAT_CONST", 15),
                ("IMPLY_NATURAL_CAT", 16), ("IMPLY_MLN_CAT", 17))
# generative models for data programming: y in {-1,1} -> index {0,1};
# labelling-function output l in {-1,0,1} -> index {0,1,2}
_DATA_PROGRAMMING = tuple(("DP_GEN_" + n, 18 + i) for i, n in enumerate((
    "CLASS_PRIOR", "LF_PRIOR", "LF_PROPENSITY", "LF_ACCURACY", "LF_CLASS_PROPENSITY",
    "DEP_FIXING", "DEP_REINFORCING", "DEP_EXCLUSIVE", "DEP_SIMILAR")))
_DISTRIBUTED = (("UFO", 30),)

FACTORS = dict(_BOOLEAN + _CATEGORICAL + _DATA_PROGRAMMING + _DISTRIBUTED)

globals().update({"FUNC_" + _name: _fid for _name, _fid in FACTORS.items()})

__all__ = ["FACTORS"] + ["FUNC_" + _name for _name in FACTORS]
