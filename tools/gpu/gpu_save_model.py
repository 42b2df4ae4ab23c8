#!/usr/bin/env python
"""Diagnostic: train a small model on the GPU, save it with VAE.save (encode.py:486-502 format) and store the latents
our encode pass produces, so that oracle/check_model_pt.py can load the file with the REFERENCE's VAE.load in the build
container and compare.   python tools/gpu/gpu_save_model.py gpurun_out/model_small.pt gpurun_out/model_small_check.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import encode as ve, synth  # noqa: E402

n, S, bs = 4096, 12, 256
ab, tnf, lens, _ = synth.features(n, S, seed=77)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
vae = ve.VAE(S, seed=3)
vae.trainmodel(dl, nepochs=4, batchsteps=[2], modelfile=sys.argv[1])
lat = vae.encode(dl)
np.savez_compressed(sys.argv[2], latent=lat, n=n, S=S, seed=77, batch=bs)
# and the round trip through our own loader
vae2 = ve.VAE.load(sys.argv[1])
assert np.array_equal(vae2.encode(dl), lat)
print("saved", sys.argv[1], "latent", lat.shape)
