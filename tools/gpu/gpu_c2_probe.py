#!/usr/bin/env python
"""Diagnostic (not a test): the whole job at a BASELINE shape for several epoch counts on ONE dataset --
train time, encode time, cluster sweep time / cluster count / purity -- to size bench.py's default workload.

    python tools/gpu/gpu_c2_probe.py N S batch precision epochs[,epochs...] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, S, bs = (int(x) for x in sys.argv[1:4])
os.environ["VAMBHIP_PRECISION"] = sys.argv[4]
epoch_list = [int(x) for x in sys.argv[5].split(",")]
out_path = sys.argv[6] if len(sys.argv) > 6 else None

from vamb_amd import cluster as vc, encode as ve, synth  # noqa: E402

t0 = time.perf_counter()
ab, tnf, lens, labels = synth.features(n, S, seed=1)
t_gen = time.perf_counter() - t0
t0 = time.perf_counter()
dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
t_prep = time.perf_counter() - t0
print(f"synthetic features {t_gen:.1f} s, make_dataloader {t_prep:.1f} s", flush=True)

results = []
for E in epoch_list:
    vae = ve.VAE(S, seed=1)
    t0 = time.perf_counter()
    vae._ensure_dataset(dl)
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    vae.trainmodel(dl, nepochs=E, batchsteps=None)
    t_train = time.perf_counter() - t0
    t0 = time.perf_counter()
    latent = vae.encode(dl)
    t_enc = time.perf_counter() - t0
    gen = vc.ClusterGenerator(latent, lens, destroy=True, rng_seed=0)
    t0 = time.perf_counter()
    ncl = 0
    pure = 0
    big = 0
    kinds = {}
    for c in gen:
        ncl += 1
        lab = labels[c.members]
        pure += int(np.bincount(lab).max())
        big += len(c.members) >= 10
        kinds[c.kind_str] = kinds.get(c.kind_str, 0) + 1
        if time.perf_counter() - t0 > 400:      # degenerate sweep: give up, report what was seen
            break
    t_clu = time.perf_counter() - t0
    gen._sync_native_counters()
    b = gen._backend
    r = dict(epochs=E, upload_s=t_up, train_s=t_train, epoch_ms=t_train / E * 1e3, encode_s=t_enc, cluster_s=t_clu,
             clusters=ncl, clusters_ge10=int(big), purity=pure / n, kinds=kinds, passes=b.scan_passes,
             medoids=b.scan_medoids, rows_streamed=b.rows_streamed, loss=vae.last_epoch_losses["loss"],
             exhausted=gen.n_remaining_points == 0)
    print(json.dumps(r), flush=True)
    results.append(r)
    b.close()
if out_path:
    with open(out_path, "w") as fh:
        json.dump(dict(n=n, S=S, batch=bs, precision=sys.argv[4], gen_s=t_gen, prep_s=t_prep, runs=results), fh, indent=1)
