#!/usr/bin/env python
"""Diagnostic (not a test): `reps` scan passes of ONE shape, for rocprofv3 --pmc runs.
    python tools/gpu/gpu_scan_one.py n L k reps [sigma]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import cluster as vc, synth  # noqa: E402
n, L, k, reps = (int(x) for x in sys.argv[1:5])
sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
lat, _ = synth.blob_latent(n, L, sigma, seed=1)
b = vc.HipScanBackend(lat, synth.lengths(n, 1).astype(np.float32), False, None)
med = np.random.RandomState(0).choice(n, k, replace=False)
for _ in range(reps):
    b.scan_raw(med)
b.close()
