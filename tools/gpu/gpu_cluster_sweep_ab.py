#!/usr/bin/env python
"""Diagnostic (not a test): ONE trained latent matrix at a BASELINE shape, clustered several times under different
generator settings (environment switches read when the generator is created) -- A/B of the sweep policy.

    python tools/gpu/gpu_cluster_sweep_ab.py N S batch precision epochs "K1=V1,K2=V2;K1=V3;..." [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, S, bs = (int(x) for x in sys.argv[1:4])
os.environ["VAMBHIP_PRECISION"] = sys.argv[4]
epochs = int(sys.argv[5])
settings = [dict(kv.split("=") for kv in s.split(",") if kv) for s in sys.argv[6].split(";")]
out_path = sys.argv[7] if len(sys.argv) > 7 else None

from vamb_amd import cluster as vc, encode as ve, synth  # noqa: E402

ab, tnf, lens, labels = synth.features(n, S, seed=1)
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
vae = ve.VAE(S, seed=1)
vae.trainmodel(dl, nepochs=epochs, batchsteps=None)
latent = vae.encode(dl)
rows = []
for st in settings:
    for k, v in st.items():
        os.environ[k] = v
    gen = vc.ClusterGenerator(latent.copy(), lens, destroy=True, rng_seed=0)
    t0 = time.perf_counter()
    ncl = 0
    npts = 0
    for c in gen:
        ncl += 1
        npts += len(c.members)
    dt = time.perf_counter() - t0
    gen._sync_native_counters()
    b = gen._backend
    r = dict(setting=st, cluster_s=dt, clusters=ncl, points=npts, passes=b.scan_passes, medoids=b.scan_medoids)
    print(json.dumps(r), flush=True)
    rows.append(r)
    b.close()
    for k in st:
        os.environ.pop(k, None)
if out_path:
    with open(out_path, "w") as fh:
        json.dump(rows, fh, indent=1)
