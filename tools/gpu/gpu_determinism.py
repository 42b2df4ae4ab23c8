#!/usr/bin/env python
"""Diagnostic (not a test): run-to-run determinism of free-running training.
    python tools/gpu/gpu_determinism.py dtype reps mode[epochs|steps] nsteps dropout"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VAMBHIP_PRECISION"] = sys.argv[1]
reps = int(sys.argv[2]); mode = sys.argv[3]; nsteps = int(sys.argv[4]); drop = float(sys.argv[5])
from vamb_amd import encode as ve, synth  # noqa: E402
B, S = 512, 6
n = B * nsteps if mode == "epochs" else 5000
ab, tnf, lens, _ = synth.features(n, S, seed=11)
sigs = []
for r in range(reps):
    dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B, destroy=True)
    vae = ve.VAE(S, dropout=drop, seed=4)
    if mode == "epochs":
        vae.trainmodel(dl, nepochs=1, batchsteps=None)
    else:
        vae._ensure_dataset(dl)
        for k in range(nsteps):
            vae.train_batch(np.arange(k * B, (k + 1) * B) % n)
    h = hashlib.sha256()
    per = {}
    for k, v in sorted(vae.state_dict().items()):
        h.update(v.numpy().tobytes())
        per[k] = hashlib.sha256(v.numpy().tobytes()).hexdigest()[:8]
    sigs.append((h.hexdigest()[:12], per))
same = all(s[0] == sigs[0][0] for s in sigs)
print(f"{sys.argv[1]} {mode} steps={nsteps} dropout={drop}:", "IDENTICAL" if same else "DIFFERENT", [s[0] for s in sigs], flush=True)
if not same:
    bad = [k for k in sigs[0][1] if any(s[1][k] != sigs[0][1][k] for s in sigs)]
    print("   tensors that differ:", bad[:40])
