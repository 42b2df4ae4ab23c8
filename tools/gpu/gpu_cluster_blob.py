#!/usr/bin/env python
"""Diagnostic (not a test): one cluster sweep over blob latents (sigma 0.08), for rocprofv3 runs of the cluster kernels.
With VAMBHIP_SCAN_DBG timing switches the results are wrong by design (the sweep is cut after 3000 clusters).
    python tools/gpu/gpu_cluster_blob.py n"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import cluster as vc, synth  # noqa: E402
n = int(sys.argv[1])
lat, _ = synth.blob_latent(n, 32, 0.08, seed=3)
lens = synth.lengths(n, 3)
t0 = time.perf_counter()
k = 0
for c in vc.ClusterGenerator(lat, lens, destroy=True, rng_seed=0):
    k += 1
    if k >= 3000:
        break
print(f"{k} clusters in {time.perf_counter() - t0:.3f} s")
