#!/usr/bin/env python
"""Diagnostic (not a test): correctness + duration of every tile / pipeline variant of the bf16 GEMM, and the cost of
the parts of the hidden-layer epilogue.   python tools/gpu/gpu_gemm16_variants.py [out.json]"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib  # noqa: E402
lib = _lib.load(); _lib.require_gpu()

def bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)

def run(epi, A, B, bias, splits, reps, variant, want=False):
    M, K = A.shape; N = B.shape[0]
    C = np.zeros((M, N), np.float32); CT = np.zeros((N, M), np.float32) if want else None
    st = np.zeros((2, N)) if want else None
    ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm16(epi, _lib.ptr(A), _lib.ptr(B), _lib.ptr(bias), _lib.ptr(C), _lib.ptr(CT), _lib.ptr(st),
                                   M, N, K, splits, reps, variant, ctypes.byref(ms)))
    return C, CT, st, ms.value

rng = np.random.RandomState(1)
out = {"correct": [], "timing": []}
# correctness of every variant on a ragged shape
M, N, K = 640, 320, 456
A = rng.standard_normal((M, K)).astype(np.float32)
B = (rng.standard_normal((N, K)) / np.sqrt(K) + 0.25 * np.arange(N)[:, None] / N).astype(np.float32)
bias = rng.standard_normal(N).astype(np.float32)
want = bf16_round(A).astype(np.float64) @ bf16_round(B).astype(np.float64).T
VARS = [0, 1, 3, 7, 21, 23, 27]   # 2x: two-buffer loop (1, 3, 7) vs three buffers + interleaved DMA issue (21, 23, 27); 0 = production
for v in VARS:
    C, _, _, _ = run(0, A, B, bias, 1, 1, v)
    e0 = float(np.abs(C - want).max() / np.abs(want).max())
    C, CT, st, _ = run(3, A, B, bias, 1, 1, v, want=True)
    z = want + bias; h = np.where(z > 0, z, 0.01 * z)
    e3 = float(np.abs(C - h).max() / np.abs(h).max())
    ok_t = bool(np.array_equal(CT, C.T))
    ok_s = bool(np.allclose(st[0], h.sum(0), rtol=1e-5, atol=1e-3))
    out["correct"].append(dict(variant=v, err_splitk=e0, err_hidden=e3, transposed_ok=ok_t, stats_ok=ok_s))
    print(f"variant {v}: split-K err {e0:.2e}  hidden err {e3:.2e}  transposed {ok_t}  stats {ok_s}", flush=True)

for (M, N, K) in [(8192, 512, 512), (8192, 512, 1120), (8192, 512, 320)]:
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
    bias = np.zeros(N, np.float32)
    for epi in (0, 3):
        for v in VARS + ([256 * 1, 256 * 2, 256 * 4, 256 * 7] if epi == 3 else []):
            _, _, _, ms = run(epi, A, B, bias, 1, 40, v)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            out["timing"].append(dict(M=M, N=N, K=K, epi=epi, variant=v, us=ms * 1e3, tflops=tf))
            print(f"{M}x{N}x{K} epi {epi} variant {v:5d}: {ms*1e3:7.2f} us {tf:7.1f} TF/s", flush=True)
# dW shape: 512x512x8192 with 8 / 16 splits
M, N, K = 512, 512, 8192
A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
for splits in (8, 16):
    for v in VARS:
        _, _, _, ms = run(0, A, B, None, splits, 40, v)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        out["timing"].append(dict(M=M, N=N, K=K, epi=0, variant=v, splits=splits, us=ms * 1e3, tflops=tf))
        print(f"dW {M}x{N}x{K}/{splits} variant {v}: {ms*1e3:7.2f} us {tf:7.1f} TF/s", flush=True)
# the same weight gradient from ROW-major operands (gemm_bf16_tn.hpp), both pipelines, tiles 64x128 (3) and 128x128 (1)
def run_tn(A, B, splits, reps, tile, pipeline):
    K, M = A.shape; N = B.shape[1]
    C = np.zeros((M, N), np.float32)
    ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm16_tn(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), None, M, N, K, K, splits, reps, tile, pipeline,
                                      ctypes.byref(ms)))
    return C, ms.value
for (M, N, K) in [(512, 512, 8192), (320, 512, 8192), (512, 32, 8192), (32, 512, 8192)]:
    A = rng.standard_normal((K, M)).astype(np.float32); B = rng.standard_normal((K, N)).astype(np.float32)
    want = bf16_round(A).astype(np.float64).T @ bf16_round(B).astype(np.float64)
    for splits in (8, 16):
        for tile in ((3, 1) if min(M, N) >= 64 else (0,)):
            for pipeline in (0, 2):
                C, ms = run_tn(A, B, splits, 40, tile, pipeline)
                err = float(np.abs(C - want).max() / np.abs(want).max())
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                out["timing"].append(dict(M=M, N=N, K=K, kind="tn", tile=tile, pipeline=pipeline, splits=splits, us=ms * 1e3, tflops=tf, err=err))
                print(f"dW row-major {M}x{N}x{K}/{splits} tile {tile} pipeline {pipeline}: {ms*1e3:7.2f} us {tf:7.1f} TF/s err {err:.1e}", flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
