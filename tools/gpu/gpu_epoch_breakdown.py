#!/usr/bin/env python
"""Where does an epoch's wall time go? (host permutation, enqueue, GPU)"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib, encode as ve, synth
lib = _lib.load()
n, S, bs = 200_000, 50, 4096
ab, tnf, lens, _ = synth.features(n, S, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
vae = ve.VAE(S, seed=1)
vae.trainmodel(dl, nepochs=2, batchsteps=None)
nb = n // bs
means = (ctypes.c_double * 5)()
t_perm = t_call = 0.0
E = 10
t0 = time.perf_counter()
for e in range(E):
    a = time.perf_counter()
    perm = np.ascontiguousarray(torch.randperm(n).numpy()[: nb * bs], dtype=np.int64)
    b = time.perf_counter()
    _lib.check(lib.vh_vae_train_epoch(vae._h, _lib.ptr(perm), nb, bs, means))
    c = time.perf_counter()
    t_perm += b - a; t_call += c - b
tot = time.perf_counter() - t0
print(f"per epoch: total {tot/E*1e3:.2f} ms  randperm {t_perm/E*1e3:.2f} ms  train_epoch call {t_call/E*1e3:.2f} ms  ({t_call/E/nb*1e6:.1f} us/step)")
t0 = time.perf_counter(); vae.trainmodel(dl, nepochs=E, batchsteps=None); print(f"trainmodel per epoch {(time.perf_counter()-t0)/E*1e3:.2f} ms")
