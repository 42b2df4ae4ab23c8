#!/usr/bin/env python
"""Step time of VAE.trainmodel under several option settings, ONE dataset, one process (A/B runs on the same box):

    python tools/gpu/gpu_step_ab.py N S batch epochs dtype "A=1;B=0|C=1|" [repeats]

Every setting is a ';'-separated list of VAMBHIP_* assignments (empty = the defaults); settings are separated by '|' and measured
round-robin `repeats` times.  Prints one line per (setting, repeat) and a summary (median us/step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
n, S, bs, E = (int(x) for x in sys.argv[1:5])
dtype = sys.argv[5]
settings = sys.argv[6].split("|") if len(sys.argv) > 6 else [""]
repeats = int(sys.argv[7]) if len(sys.argv) > 7 else 2
os.environ["VAMBHIP_PRECISION"] = dtype
from vamb_amd import encode as ve, synth
ab, tnf, lens, _ = synth.features(n, S, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
steps = n // bs
res = {s: [] for s in settings}
touched = set()
for rep in range(repeats):
    for s in settings:
        for k in touched:
            os.environ.pop(k, None)
        for kv in filter(None, s.split(";")):
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        vae = ve.VAE(S, seed=1)
        vae.trainmodel(dl, nepochs=2, batchsteps=None)
        t0 = time.perf_counter()
        vae.trainmodel(dl, nepochs=E, batchsteps=None)
        us = (time.perf_counter() - t0) / E / steps * 1e6
        res[s].append(us)
        print(f"{dtype} N={n} S={S} batch={bs} [{s or 'defaults'}] rep {rep}: {us:.1f} us/step, loss {vae.last_epoch_losses['loss']:.4f}", flush=True)
        del vae
for s in settings:
    print(f"SUMMARY [{s or 'defaults'}]: median {np.median(res[s]):.1f} us/step, min {min(res[s]):.1f}, all {['%.1f' % x for x in res[s]]}")
