#!/usr/bin/env python
"""Diagnostic (not a test): duration / TFLOP/s of the bf16-storage GEMM at the shapes of the C2 / C3 training step.
    python tools/gpu/gpu_gemm16_bench.py [out.json]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
rows = []
shapes = [  # (what, epi, M, N, K, splits)
    ("C3 enc0 fwd (K=D)", 3, 8192, 512, 1120, 1), ("C2 enc0 fwd (K=D)", 3, 8192, 512, 320, 1),
    ("hidden fwd 512", 3, 8192, 512, 512, 1), ("hidden fwd 512, M=16384", 3, 16384, 512, 512, 1),
    ("out fwd C2", 1, 8192, 320, 512, 1), ("out fwd C3", 1, 8192, 1120, 512, 1),
    ("mu fwd", 0, 8192, 32, 512, 4), ("dW 512x512", 0, 512, 512, 8192, 8), ("dW 512x512 x16", 0, 512, 512, 8192, 16),
    ("dW enc0 C3", 0, 512, 1120, 8192, 4), ("dW mu", 0, 32, 512, 8192, 16), ("square 4096", 0, 4096, 4096, 4096, 1),
]
rng = np.random.RandomState(0)
for what, epi, M, N, K, splits in shapes:
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    bias = np.zeros(N, np.float32)
    C = np.zeros((M, N), np.float32)
    ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm16(epi, _lib.ptr(A), _lib.ptr(B), _lib.ptr(bias), _lib.ptr(C), None, None, M, N, K, splits, 50, 0,
                                   ctypes.byref(ms)))
    tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
    rows.append(dict(what=what, epi=epi, M=M, N=N, K=K, splits=splits, us=ms.value * 1e3, tflops=tf, frac_bf16_peak=tf / 2500.0))
    print(f"{what:28s} epi {epi} {M}x{N}x{K} /{splits}: {ms.value*1e3:8.2f} us  {tf:8.1f} TF/s ({tf/25:.1f} % of 2.5 PF)", flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump(rows, fh, indent=1)
