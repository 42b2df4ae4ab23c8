#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (oracle/; build container): load a model.pt written by vamb_amd.encode.VAE.save on the GPU
box with the REFERENCE's own VAE.load (vamb/encode.py:504-541, torch.load(weights_only=True) + load_state_dict) and
compare the reference's encode() on the same inputs with the latents the GPU produced.

    python oracle/check_model_pt.py gpurun_out/model_small.pt gpurun_out/model_small_check.npz [out.json]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from vamb_amd import synth  # noqa: E402

vt, rc, re_ = ref_harness.load_reference()
chk = np.load(sys.argv[2])
n, S, seed, bs = int(chk["n"]), int(chk["S"]), int(chk["seed"]), int(chk["batch"])
ab, tnf, lens, _ = synth.features(n, S, seed=seed)
dl = re_.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True, cuda=False)
vae = re_.VAE.load(sys.argv[1], cuda=False, evaluate=True)       # the reference's loader: strict load_state_dict
ref_lat = vae.encode(dl)
ours = chk["latent"]
err = float(np.abs(ref_lat - ours).max() / np.abs(ref_lat).max())
keys = list(vae.state_dict().keys())
out = dict(model=os.path.basename(sys.argv[1]), loaded_by="reference vamb.encode.VAE.load (strict state_dict)",
           n_state_entries=len(keys), latent_shape=list(ref_lat.shape), max_rel_latent_error=err,
           identical_masked_latents=float(np.mean(ref_lat == ours)))
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
assert err < 2.0 ** -10, err
