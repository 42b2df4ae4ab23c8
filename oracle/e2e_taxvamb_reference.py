"""TEST INFRASTRUCTURE ONLY -- free-running joint TaxVamb training of the REAL reference.

``make_dataloader_*_hloss -> VAEVAEHLoss.trainmodel -> VAEJoint.encode`` through ``oracle/ref_harness.py`` on the synthetic
taxonomy problem of tests/golden/fixture_defs.py (taxvamb_problem), torch's own RNG.  Records the 17 metrics of every epoch's log
line (semisupervised_encode.py:1001-1006) and how well the joint latent separates the leaves.  Build container only.

    python oracle/e2e_taxvamb_reference.py            # every model seed of fixture_defs.TAXVAMB_E2E -> tests/golden/taxvamb_e2e_reference.json
"""
from __future__ import annotations

import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

_PAIR_RE = re.compile(r"(\w+): (\S+)")


class EpochLog:
    def __init__(self):
        self.rows = []

    def info(self, msg, *a, **k):
        msg = str(msg)
        if "Epoch:" in msg and "loss_vamb" in msg:
            self.rows.append({k: float(v) for k, v in _PAIR_RE.findall(msg) if k != "Epoch"})
        return self

    def __getattr__(self, name):
        return lambda *a, **k: self


def leaf_separation(latent, nodes):
    """mean distance of a leaf's contigs to their centroid / mean distance between the centroids of different leaves"""
    rows = [np.flatnonzero(nodes == k) for k in range(5, 14)]
    cent = np.stack([latent[r].mean(axis=0) for r in rows])
    own = np.mean([np.linalg.norm(latent[r] - cent[i], axis=1).mean() for i, r in enumerate(rows)])
    other = np.mean([np.linalg.norm(cent[i] - cent[j]) for i in range(9) for j in range(9) if i != j])
    return float(own / other)


def run_reference(c, model_seed, threads=8):
    import torch

    import fixture_defs as fd
    import ref_harness

    _, _, en = ref_harness.load_reference()
    ss = ref_harness.load_reference_module("semisupervised_encode")
    tx = ref_harness.load_reference_module("taxvamb_encode")
    torch.set_num_threads(threads)
    ab, tnf, lens, nodes, parents = fd.taxvamb_problem(c["n"], c["nsamples"], c["data_seed"])
    N, B, S = len(parents), c["batch"], c["nsamples"]
    names = [f"n{i}" for i in range(N)]
    dl_v = en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B)
    dl_j = tx.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B)
    dl_l = tx.make_dataloader_labels_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B)
    dl = tx.make_dataloader_semisupervised_hloss(dl_j, dl_v, dl_l, N, parents, (S, 103, 1, N), c["perm_seed"], batchsize=B)
    torch.manual_seed(model_seed)
    vae = tx.VAEVAEHLoss(S, N, names, parents, nhiddens=list(c["nhiddens"]), nlatent=c["nlatent"])
    torch.manual_seed(model_seed + 1000)   # (VAE.__init__ reseeds with its own default seed 0: make the runs differ)
    for net in (vae.VAEVamb, vae.VAELabels, vae.VAEJoint):
        for m in net.modules():
            if isinstance(m, torch.nn.Linear):
                m.reset_parameters()
    log = EpochLog()
    saved = ss.logger
    ss.logger = log
    try:
        t0 = time.perf_counter()
        vae.trainmodel(dl, nepochs=c["nepochs"], batchsteps=list(c["batchsteps"]))
        t_train = time.perf_counter() - t0
    finally:
        ss.logger = saved
    latent = vae.VAEJoint.encode(dl_j)
    return dict(model_seed=model_seed, t_train=t_train, epochs=log.rows, leaf_separation=leaf_separation(latent, nodes))


if __name__ == "__main__":
    import fixture_defs as fd

    c = fd.TAXVAMB_E2E
    runs = [run_reference(c, seed) for seed in c["model_seeds"]]
    out = dict(config={k: v for k, v in c.items()}, runs=runs,
               note="real reference (RasmussenLab/vamb taxvamb_encode.VAEVAEHLoss.trainmodel), CPU, torch RNG: a SPREAD to land in")
    path = os.path.join(ROOT, "tests", "golden", "taxvamb_e2e_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    for r in runs:
        print(r["model_seed"], round(r["t_train"], 2), "s", {k: round(r["epochs"][-1][k], 5) for k in ("loss", "loss_joint", "ce_labels_joint", "loss_vamb", "loss_labels")},
              "first loss", round(r["epochs"][0]["loss"], 4), "sep", round(r["leaf_separation"], 3))
