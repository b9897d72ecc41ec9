#!/usr/bin/env python3
"""Known-byte-count workload for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950:
each k_stream_copy launch reads and writes exactly NBYTES."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numbskull_amd import _lib

NBYTES = 1 << 30
for width in (4, 16):
    g = C.c_double()
    _lib.check(_lib.lib().nsk_selftest_stream(0, NBYTES, width, 3, C.byref(g)))
    print("stream copy width=%d: %.1f GB/s (read+write), %d bytes each way per launch"
          % (width, g.value, NBYTES))
