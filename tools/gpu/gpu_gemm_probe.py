#!/usr/bin/env python
"""Run one GEMM instantiation repeatedly (for rocprofv3 --pmc / --kernel-trace runs)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib
lib = _lib.load()
tile, a_kc, b_kc, M, N, K, splits, reps = [int(x) for x in sys.argv[1:9]]
rng = np.random.RandomState(0)
A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
Ad = np.ascontiguousarray(A if a_kc else A.T); Bd = np.ascontiguousarray(B if b_kc else B.T)
C = np.zeros((M, N), np.float32); ms = ctypes.c_float()
for _ in range(reps):
    _lib.check(lib.vh_debug_gemm(tile, a_kc, b_kc, _lib.ptr(Ad), _lib.ptr(Bd), None, _lib.ptr(C), M, N, K, splits, ctypes.byref(ms)))
print("ms", ms.value, "TF", 2.0 * M * N * K / ms.value / 1e9)
