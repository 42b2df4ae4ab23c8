"""Diagnostic: vh_debug_gemm against float64 numpy at large shapes: tile a_kc b_kc M N K splits"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from vamb_amd import _lib
lib = _lib.load()
for spec in sys.argv[1:]:
    tile, a_kc, b_kc, M, N, K, splits = [int(x) for x in spec.split(",")]
    rng = np.random.RandomState(0)
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    Ad = np.ascontiguousarray(A if a_kc else A.T); Bd = np.ascontiguousarray(B if b_kc else B.T)
    C = np.zeros((M, N), np.float32); ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm(tile, a_kc, b_kc, _lib.ptr(Ad), _lib.ptr(Bd), None, _lib.ptr(C), M, N, K, splits, ctypes.byref(ms)))
    err = np.abs(C - want)
    bad = np.argwhere(err > 1e-3 * np.abs(want).max())
    print(spec, "max err", err.max(), "rel", err.max() / np.abs(want).max(), "bad elements", len(bad), bad[:5].tolist())
