"""Diagnostic (not a test): cProfile of the host side of the cluster sweep on C1-shaped latents.
Usage on the GPU box: python tools/gpu/gpu_cluster_profile.py [contigs] [epochs]"""
import cProfile
import io
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from vamb_amd import cluster as vc, encode as ve, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ab, tnf, lens, _ = synth.features(n, 50, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=4096, destroy=True)
vae = ve.VAE(50, nlatent=32, seed=0)
vae.trainmodel(dl, nepochs=epochs, batchsteps=None)
latent = vae.encode(dl)
np.save("gpurun_out/c1_latent.npy", latent[:0])  # keep gpurun_out present

for mode in ("plain", "timing", "profile"):
    gen = vc.ClusterGenerator(latent.copy(), lens, destroy=True, rng_seed=0)
    gen._backend.set_timing(mode == "timing")
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if mode == "profile":
        pr.enable()
    ncl = sum(1 for _ in gen)
    if mode == "profile":
        pr.disable()
    dt = time.perf_counter() - t0
    b = gen._backend
    print(f"[{mode}] sweep {dt*1e3:.1f} ms, {ncl} clusters, passes {b.scan_passes}, medoids {b.scan_medoids}, "
          f"kernel {b.kernel_ms:.1f} ms")
    if mode == "profile":
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
        print(s.getvalue())
