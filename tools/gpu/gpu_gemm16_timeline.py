#!/usr/bin/env python
"""Diagnostic (not a test): where the time of one bf16 GEMM launch goes, from in-kernel s_memtime stamps of every workgroup.
    python tools/gpu/gpu_gemm16_timeline.py [out.txt]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib  # noqa: E402
lib = _lib.load(); _lib.require_gpu()

def timeline(epi, M, N, K, variant):
    cap = 4096
    st = np.zeros((cap, 8), np.uint64)
    nb = ctypes.c_int(); ms = ctypes.c_float()
    _lib.check(lib.vh_debug_gemm16_timeline(epi, M, N, K, variant, _lib.ptr(st), cap, ctypes.byref(nb), ctypes.byref(ms)))
    st = st[:nb.value].astype(np.float64)
    return st, ms.value * 1e3

lines = []
for (M, N, K) in [(8192, 512, 320), (8192, 512, 512), (8192, 512, 1120)]:
    for epi in (0, 3):
        for variant in (1, 21):
            st, us = timeline(epi, M, N, K, variant)
            t0 = st[:, 0].min()
            rel = st[:, :5] - t0
            rel[:, 3] = np.where(st[:, 3] == 0, rel[:, 2], rel[:, 3])   # epilogues without an LDS image have no phase 1
            span = rel[:, 4].max()
            # ticks -> us: the whole launch (first entry .. last 'stores issued') is a bit shorter than the event time
            names = ["entry", "tile0 landed", "K loop done", "phase 1 done", "stores issued"]
            seg = np.diff(rel, axis=1)
            l = (f"{M}x{N}x{K} epi {epi} variant {variant:2d}: {us:6.2f} us/launch, {len(st)} workgroups, span {span:.0f} ticks; "
                 f"entry spread {rel[:,0].max():.0f}; mean ticks per segment: " +
                 ", ".join(f"{n} {seg[:, i][seg[:, i] < 1e9].mean():.0f}" for i, n in enumerate(["prologue", "K loop", "phase 1", "stores"])) +
                 f"; ticks/us (span-based) {span / us:.1f}")
            print(l, flush=True); lines.append(l)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
