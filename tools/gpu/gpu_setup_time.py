"""Diagnostic: host-side fixed costs of one job at C1 (VAE construction, dataset residency check, cluster set-up)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from vamb_amd import cluster as vc, encode as ve, synth
n = 200000
ab, tnf, lens, _ = synth.features(n, 50, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=4096, destroy=True)
for rep in range(3):
    t0 = time.perf_counter(); vae = ve.VAE(50, nlatent=32, seed=rep); t1 = time.perf_counter()
    vae._ensure_dataset(dl); t2 = time.perf_counter()
    vae.trainmodel(dl, nepochs=1, batchsteps=None); t3 = time.perf_counter()
    lat = vae.encode(dl); t4 = time.perf_counter()
    gen = vc.ClusterGenerator(lat, lens, destroy=True, rng_seed=rep); t5 = time.perf_counter()
    c = next(gen); t6 = time.perf_counter()
    print(f"VAE() {1e3*(t1-t0):.1f} ms, ensure_dataset {1e3*(t2-t1):.1f}, 1 epoch {1e3*(t3-t2):.1f}, encode {1e3*(t4-t3):.1f}, "
          f"ClusterGenerator() {1e3*(t5-t4):.1f}, first cluster {1e3*(t6-t5):.1f}")
