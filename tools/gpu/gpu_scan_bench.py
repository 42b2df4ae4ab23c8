#!/usr/bin/env python
"""Diagnostic (not a test): duration of one scan pass by matrix size and medoid count, unrolled-load kernel vs runtime-width kernel
(VAMBHIP_SCAN_LC=1 / 0).   python tools/gpu/gpu_scan_bench.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import cluster as vc, synth  # noqa: E402

mode = os.environ.get("VAMBHIP_SCAN_LC", "1")
rows = []
for (n, L) in [(100_000, 32), (400_000, 32), (2_000_000, 32), (2_000_000, 64)]:
    lat, _ = synth.blob_latent(n, L, 0.3, seed=1)      # sigma 0.3: a few percent of the pairs fall inside the histogram range
    lens = synth.lengths(n, 1)
    b = vc.HipScanBackend(lat, lens.astype(np.float32), False, None)
    rng = np.random.RandomState(0)
    for k in (1, 4, 8, 9, 16, 25, 32):
        med = rng.choice(n, k, replace=False)
        b.scan_raw(med)          # warm-up
        b.set_timing(True)
        b.kernel_ms = 0.0
        reps = 10
        for _ in range(reps):
            b.scan_raw(med)
        kms = b.kernel_ms / reps
        b.set_timing(False)
        t0 = time.perf_counter()
        for _ in range(50):
            b.scan_raw(med)
        wall = (time.perf_counter() - t0) / 50
        gbs = n * (4 * ((L + 3) // 4 * 4) + 5) / (kms * 1e-3) / 1e9
        rows.append(dict(n=n, L=L, k=k, lc=mode, kernel_us=kms * 1e3, wall_us=wall * 1e6, gbps=gbs))
        print(f"lc={mode} n={n} L={L} k={k:2d}: kernel {kms*1e3:7.1f} us  pass wall {wall*1e6:7.1f} us  {gbs:7.0f} GB/s", flush=True)
    b.close()
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
