#!/usr/bin/env python
"""Diagnostic (not a test): `reps` launches of ONE bf16 GEMM shape through vh_debug_gemm16, for rocprofv3 --pmc runs.
    python tools/gpu/gpu_gemm16_one.py epi M N K reps [t|-] [variant]     (variant: vh_debug_gemm16's, e.g. 24 + 256 * 32 = the
    128 x 64 two-per-CU tile with the step's hashed dropout)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib  # noqa: E402
lib = _lib.load(); _lib.require_gpu()
epi, M, N, K, reps = (int(x) for x in sys.argv[1:6])
rng = np.random.RandomState(0)
A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
bias = np.zeros(N, np.float32); C = np.zeros((M, N), np.float32)
CT = np.zeros((N, M), np.float32); st = np.zeros((2, N))
ms = ctypes.c_float()
want_t = len(sys.argv) > 6 and sys.argv[6] == "t"   # "t": also the transposed copy of the round-2 dataflow (general epilogue)
_lib.check(lib.vh_debug_gemm16(epi, _lib.ptr(A), _lib.ptr(B), _lib.ptr(bias), _lib.ptr(C), _lib.ptr(CT) if want_t else None, _lib.ptr(st), M, N, K, 1, reps,
                               int(sys.argv[7]) if len(sys.argv) > 7 else 0, ctypes.byref(ms)))
print(f"epi {epi} {M}x{N}x{K}: {ms.value*1e3:.2f} us per launch, {2.0*M*N*K/(ms.value*1e-3)/1e12:.1f} TF/s")
