#!/usr/bin/env python
"""Per-epoch wall time of vh_vae_train_epoch over many epochs (looking for sporadic stalls)."""
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
perm = np.ascontiguousarray(torch.randperm(n).numpy()[: nb * bs], dtype=np.int64)
ts = []
for e in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    a = time.perf_counter()
    _lib.check(lib.vh_vae_train_epoch(vae._h, _lib.ptr(perm), nb, bs, means))
    ts.append((time.perf_counter() - a) * 1e3)
ts = np.array(ts)
print("epoch ms: min %.2f median %.2f mean %.2f max %.2f" % (ts.min(), np.median(ts), ts.mean(), ts.max()))
print(np.round(ts, 1).tolist())
