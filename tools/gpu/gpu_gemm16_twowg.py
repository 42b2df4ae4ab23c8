#!/usr/bin/env python
"""Diagnostic (not a test): the batch-tall training GEMM (E16_HIDDEN_TRAIN, hashed dropout as in the step) at the three
BASELINE shapes under every tile that puts ONE (1, 21) or TWO (23, 24, 28: 72 KB of LDS each) workgroups on a CU -- VERDICT r5
item 3(a).   python tools/gpu/gpu_gemm16_twowg.py [out.json]"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib  # noqa: E402
lib = _lib.load(); _lib.require_gpu()

def run(epi, A, B, bias, reps, variant):
    M, K = A.shape; N = B.shape[0]
    C = np.zeros((M, N), np.float32)
    ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm16(epi, _lib.ptr(A), _lib.ptr(B), _lib.ptr(bias), _lib.ptr(C), None, None, M, N, K, 1, reps, variant,
                                   ctypes.byref(ms)))
    return C, ms.value

def bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)

rng = np.random.RandomState(2)
rows = []
# correctness of the new tiles (no dropout) on a ragged shape and on a BASELINE shape
for (M, N, K) in [(640, 320, 456), (8192, 512, 320)]:
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    z = bf16_round(A).astype(np.float64) @ bf16_round(B).astype(np.float64).T + bias
    h = np.where(z > 0, z, 0.01 * z)
    for v in (21, 23, 24, 28):
        C, _ = run(3, A, B, bias, 1, v)
        err = float(np.abs(C - h).max() / np.abs(h).max())
        print(f"correct {M}x{N}x{K} variant {v}: max rel err {err:.2e}", flush=True)
        assert err < 1e-2, (v, err)
DROP = 256 * 32
for (M, N, K) in [(8192, 512, 320), (8192, 512, 512), (8192, 512, 1120)]:
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
    bias = np.zeros(N, np.float32)
    for rep in range(2):
        for v in (21, 1, 23, 24, 28):
            _, ms = run(3, A, B, bias, 200, v + DROP)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            rows.append(dict(M=M, N=N, K=K, variant=v, rep=rep, us=ms * 1e3, tflops=tf))
            print(f"{M}x{N}x{K} hidden-train + dropout, variant {v:3d}: {ms*1e3:7.2f} us {tf:7.1f} TF/s", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
