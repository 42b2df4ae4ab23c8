#!/usr/bin/env python
"""What the roofline probe costs the step it measures: C2 training epochs with the probe off, timing every launch of the
layer-0 GEMM, and timing every 16th (the default).   python tools/gpu/gpu_probe_ab.py [epochs]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 40
os.environ["VAMBHIP_PRECISION"] = "bf16"
from vamb_amd import _lib, encode as ve, synth
n, S, bs = 2_000_000, 200, 8192
ab, tnf, lens, _ = synth.features(n, S, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=False)
lib = _lib.load()
for label, on, every in (("probe off", 0, 16), ("every launch", 1, 1), ("every 16th", 1, 16), ("probe off", 0, 16), ("every 16th", 1, 16)):
    _lib.set_option("vae.probe_every", every)
    vae = ve.VAE(S, seed=1)
    _lib.check(lib.vh_vae_set_probe(vae._h, on, 0))
    vae.trainmodel(dl, nepochs=2, batchsteps=None)
    t0 = time.perf_counter()
    vae.trainmodel(dl, nepochs=E, batchsteps=None)
    dt = (time.perf_counter() - t0) / E
    ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(lib.vh_vae_probe_result(vae._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
    per = ms.value / nl.value * 1e3 if nl.value else float("nan")
    print(f"{label:14s}: {dt * 1e3:7.2f} ms/epoch, {dt / (n // bs) * 1e6:6.1f} us/step; probed launches {nl.value}, avg {per:.2f} us", flush=True)
