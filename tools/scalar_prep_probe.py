#!/usr/bin/env python3
"""Cost of the scalar preparation on its own: k_debug_half_scalars (halfgcd.h) and k_debug_lattice3
(lattice3.h) on 2^LOG2N random challenges; run under `rocprofv3 --kernel-trace` and read the kernel
durations (the host call includes the copies)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401

from schnorr_amd import _lib  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402

E.init(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
rng = np.random.default_rng(3)
c = rng.integers(0, 256, (n, 32), dtype=np.uint8)
c[:, 31] &= 3
u = rng.integers(0, 256, (n, 32), dtype=np.uint8)
u[:, 31] &= 7
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
o96, o128 = np.zeros((n, 96), np.uint8), np.zeros((n, 128), np.uint8)
L = _lib.load()
for _ in range(5):
    _lib.check(L.dsv_debug_half_scalars(p(c), ctypes.c_size_t(n), p(o96)))
    _lib.check(L.dsv_debug_lattice3(p(u), p(c), ctypes.c_size_t(n), p(o128)))
print("done")
