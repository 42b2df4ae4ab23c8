"""Diagnostic (GPU box): the fp32 GEMM at the joint TaxVamb step's shapes (batch 256: a few dozen workgroups, each alone on its CU).
Times vh_debug_gemm (50 back-to-back launches) for the tile variants -- 3 = the production 64 x 64 tile, 4 = 64-wide K-tiles,
5 = one free-running wavefront per 32 x 32 tile, 6 = 64 x 64 with four K-tiles in flight (gemm.hpp, PF = 4), 7 = 32 x 32 with four
wavefront groups over K (KS = 4) -- and checks each
result against float64 numpy.

    python tools/gpu/gpu_gemm_small.py [out.txt]
"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from vamb_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.set_option("debug.gemm_reps", 50)
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def emit(line):
    print(line, flush=True)
    if out:
        out.write(line + "\n")


# (a_kc, b_kc, M, N, K, splits): forward / dX (K-contiguous A; B K-contiguous or row-contiguous), dW (both row-contiguous, K = batch)
CASES = [(1, 1, 256, 512, 512, 1), (1, 0, 256, 512, 512, 1), (1, 1, 256, 512, 1024, 1), (1, 1, 256, 1152, 512, 1),
         (1, 1, 512, 512, 512, 1), (1, 1, 1024, 512, 512, 1), (1, 1, 4096, 512, 512, 1), (0, 0, 512, 512, 256, 1), (0, 0, 512, 512, 256, 2),
         (0, 0, 512, 512, 4096, 8), (1, 1, 256, 512, 512, 2), (1, 1, 256, 512, 512, 4)]
for a_kc, b_kc, M, N, K, splits in CASES:
    rng = np.random.RandomState(0)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    Ad = np.ascontiguousarray(A if a_kc else A.T)
    Bd = np.ascontiguousarray(B if b_kc else B.T)
    row = []
    for tile in (3, 4, 5, 6, 7, 2):
        if tile == 4 and (K // splits) % 64:
            continue
        if tile == 6 and ((K // splits) % 128 or K % splits):
            continue
        if tile == 7 and ((K // splits) % 512 or K % splits):
            continue
        C = np.zeros((M, N), np.float32)
        ms = ctypes.c_float()
        _lib.check(lib.vh_debug_gemm(tile, a_kc, b_kc, _lib.ptr(Ad), _lib.ptr(Bd), None, _lib.ptr(C), M, N, K, splits, ctypes.byref(ms)))
        err = np.abs(C - want).max() / np.abs(want).max()
        row.append(f"tile {tile}: {1e3 * ms.value:6.2f} us (err {err:.1e})")
    emit(f"A {'KC' if a_kc else 'RC'} B {'KC' if b_kc else 'RC'} M={M:5d} N={N:5d} K={K:5d} splits={splits}: " + "  ".join(row))
