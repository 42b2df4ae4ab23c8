"""Diagnostic: scan kernel time on C1-like latents (trained VAE) per medoid count; VAMBHIP_SCAN_DBG toggles parts."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from vamb_amd import cluster as vc, encode as ve, synth
n = 200000
ab, tnf, lens, _ = synth.features(n, 50, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=4096, destroy=True)
vae = ve.VAE(50, nlatent=32, seed=0)
vae.trainmodel(dl, nepochs=60, batchsteps=None)
lat = vae.encode(dl)
b = vc.HipScanBackend(lat, lens.astype(np.float32), False, None)
b.set_timing(True)
rng = np.random.RandomState(0)
out = []
for k in (1, 8, 16, 25):
    meds = [int(x) for x in rng.choice(n, size=k, replace=False)]
    st = b.scan(meds)
    b.kernel_ms = 0.0
    for _ in range(20):
        b.scan(meds)
    frac = float(np.mean([s.hist_fx.sum() for s in st])) / 256.0 / float(lens.sum())
    out.append(f"k={k}: {b.kernel_ms/20*1e3:.1f} us (hist mass fraction {frac:.3f})")
print("dbg", os.environ.get("VAMBHIP_SCAN_DBG", "0"), " | ".join(out))
