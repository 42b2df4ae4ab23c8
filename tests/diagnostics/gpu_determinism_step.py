#!/usr/bin/env python
"""Diagnostic (not a test): ONE free-running training step on identical fresh models, repeated; which tensors differ between
repetitions?   python tests/diagnostics/gpu_determinism_step.py dtype reps [batch] [nsamples]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["VAMBHIP_PRECISION"] = sys.argv[1]
reps = int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 512; S = int(sys.argv[4]) if len(sys.argv) > 4 else 6
import vae_oracle as vo
from vamb_amd import encode as ve, synth
ab, tnf, lens, _ = synth.features(max(B, 2048), S, seed=11)
names = vo.param_names([512, 512])
runs = []
for r in range(reps):
    dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B, destroy=True)
    vae = ve.VAE(S, seed=4)
    vae._ensure_dataset(dl)
    losses = vae.train_batch(np.arange(B))
    rec = {"losses": np.array(losses)}
    for li in range(4):
        rec[f"hidden{li}"] = vae.hidden_activations(li, B)
    for n in names:
        rec["grad:" + n] = vae.parameters_gradient(n)
    runs.append(rec)
for k in runs[0]:
    diffs = [int((runs[r][k] != runs[0][k]).sum()) for r in range(1, reps)]
    mx = [float(np.abs(runs[r][k].astype(np.float64) - runs[0][k]).max()) for r in range(1, reps)]
    print(f"{k:32s} differing elements vs run 0: {diffs}  max abs diff {['%.2e' % m for m in mx]}  (|x|max {np.abs(runs[0][k]).max():.3e})")
