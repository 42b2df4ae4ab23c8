"""Diagnostic (GPU box): throughput of the joint TaxVamb trainer (vamb_amd.taxvamb_encode.VAEVAEHLoss.trainmodel) on a synthetic
problem -- n contigs, S samples, a random taxonomy of N nodes -- at the CLI's starting batch size and at a large batch.

    python tools/gpu/gpu_taxvamb_bench.py N_CONTIGS S N_NODES [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import encode as ve, synth, taxvamb_encode as vt  # noqa: E402


def random_tree(n_nodes, seed=0):
    rng = np.random.RandomState(seed)
    parents = [-1]
    for i in range(1, n_nodes):
        parents.append(int(rng.randint(max(0, i - 60), i)))
    return parents


def run(n, S, n_nodes):
    ab, tnf, lens, genome = synth.features(n, S, seed=3)
    parents = random_tree(n_nodes)
    nodes = (genome.astype(np.int64) * 7919) % n_nodes
    names = [f"n{i}" for i in range(n_nodes)]
    out = dict(n=n, S=S, n_nodes=n_nodes, runs=[])
    for B in (256, 4096):
        dl_v = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B)
        dl_j = vt.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, nodes, n_nodes, parents, batchsize=B)
        dl_l = vt.make_dataloader_labels_hloss(ab.copy(), tnf.copy(), lens, nodes, n_nodes, parents, batchsize=B)
        dl = vt.make_dataloader_semisupervised_hloss(dl_j, dl_v, dl_l, n_nodes, parents, (S, 103, 1, n_nodes), 0, batchsize=B)
        vae = vt.VAEVAEHLoss(S, n_nodes, names, parents)
        vae.trainmodel(dl, nepochs=1, batchsteps=None)            # warm-up epoch (uploads, first launches)
        t0 = time.perf_counter()
        vae.trainmodel(dl, nepochs=2, batchsteps=None)
        dt = (time.perf_counter() - t0) / 2
        steps = n // B
        t0 = time.perf_counter()
        lat = vae.VAEJoint.encode(dl_j)
        t_enc = time.perf_counter() - t0
        out["runs"].append(dict(batch=B, steps_per_epoch=steps, s_per_epoch=dt, ms_per_step=1e3 * dt / steps, contigs_per_s=steps * B / dt,
                                encode_s=t_enc, loss=vae.last_epoch_metrics["loss"], latent_finite=bool(np.isfinite(lat).all())))
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    res = run(int(a[0]), int(a[1]), int(a[2]))
    line = json.dumps(res)
    print(line)
    if len(a) > 3:
        with open(a[3], "a") as fh:
            fh.write(line + "\n")
