"""Identity stand-in for numba, used ONLY by tools/make_goldens.py to import the
reference in its sanctioned pure-Python mode (the reference's CI runs with
NUMBA_DISABLE_JIT=1).  Not part of the product."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn
