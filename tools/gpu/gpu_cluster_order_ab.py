#!/usr/bin/env python
"""Diagnostic (not a test): the same blob latents (sigma 0.5: many small clusters, a pass-bound sweep) clustered with the tuned
kernels (ascending chain) and in the reference's evaluation order (scan.reference_order: the plain kernel).
    python tools/gpu/gpu_cluster_order_ab.py n [max_clusters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import cluster as vc, synth  # noqa: E402
n = int(sys.argv[1])
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 60
lat, _ = synth.blob_latent(n, 32, 0.5, seed=3)
lens = synth.lengths(n, 3)
for mode in ("0", "2", "1"):
    os.environ["VAMBHIP_REFERENCE_ORDER"] = mode
    gen = vc.ClusterGenerator(lat.copy(), lens, destroy=True, rng_seed=0)
    t0 = time.perf_counter()
    k = pts = 0
    for c in gen:
        k += 1
        pts += len(c.members)
        if k >= cap:
            break
    dt = time.perf_counter() - t0
    gen._sync_native_counters()
    b = gen._backend
    print(f"reference_order={mode}: {k} clusters, {pts} points in {dt:.3f} s, {b.scan_passes} passes, {dt / max(1, b.scan_passes) * 1e6:.1f} us per pass", flush=True)
    b.close()
