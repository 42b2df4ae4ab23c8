#!/usr/bin/env python
"""Epoch time of VAE.trainmodel at a given shape / precision: python tools/gpu/gpu_epoch_time.py N S batch epochs [fp32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, S, bs, E = (int(x) for x in sys.argv[1:5])
os.environ["VAMBHIP_PRECISION"] = sys.argv[5] if len(sys.argv) > 5 else "fp32"
from vamb_amd import encode as ve, synth
ab, tnf, lens, _ = synth.features(n, S, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
vae = ve.VAE(S, seed=int(os.environ.get("VAE_SEED", "1")))
vae.trainmodel(dl, nepochs=2, batchsteps=None)
t0 = time.perf_counter()
vae.trainmodel(dl, nepochs=E, batchsteps=None)
dt = (time.perf_counter() - t0) / E
steps = n // bs
D = S + 104
flops = 12 * 512 * (D + 512 + 32) - 2 * D * 512
t1 = time.perf_counter(); lat = vae.encode(dl); te = time.perf_counter() - t1
print(f"{os.environ['VAMBHIP_PRECISION']} N={n} S={S} batch={bs}: {dt*1e3:.2f} ms/epoch, {dt/steps*1e6:.1f} us/step, "
      f"{n/dt/1e6:.2f} M contigs/s/epoch, {flops*n/dt/1e12:.1f} TFLOP/s algorithmic, encode {te*1e3:.1f} ms, "
      f"loss {vae.last_epoch_losses['loss']:.4f}")
