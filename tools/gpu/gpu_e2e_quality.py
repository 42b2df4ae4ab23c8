"""Diagnostic (GPU box): free-running training -> encode -> clustering through the product path on synthetic features,
with the agreement of the bins with the synthetic genomes (tests/golden/fixture_defs.bin_quality) and the loss curve.

    python tools/gpu/gpu_e2e_quality.py N S nepochs batchsize '[batchsteps]' dtype model_seed [data_seed] [out.json]
"""
import json
import logging
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import fixture_defs as fd  # noqa: E402
from vamb_amd import cluster as vc, encode as ve, synth  # noqa: E402

_EPOCH_RE = re.compile(r"Epoch:\s*(\d+)\s+Loss:\s*(\S+)\s+CE:\s*(\S+)\s+AB:\s*(\S+)\s+SSE:\s*(\S+)\s+KLD:\s*(\S+)\s+Batchsize:\s*(\d+)")


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.rows = []

    def emit(self, record):
        m = _EPOCH_RE.search(record.getMessage())
        if m:
            self.rows.append([float(m.group(i)) for i in (2, 3, 4, 5, 6, 7)])


def run(n, S, nepochs, bs, steps, dtype, seed, dseed=1):
    ab, tnf, lens, labels = synth.features(n, S, seed=dseed)
    previous = ve._COMPUTE_DTYPE
    ve.set_compute_dtype(dtype)
    cap = Capture()
    log = logging.getLogger("vamb_amd.encode")
    log.setLevel(logging.INFO)
    log.addHandler(cap)
    try:
        dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs)
        vae = ve.VAE(S, seed=seed)
        t0 = time.perf_counter()
        vae.trainmodel(dl, nepochs=nepochs, batchsteps=steps)
        t_train = time.perf_counter() - t0
        latent = vae.encode(dl)
    finally:
        log.removeHandler(cap)
        ve.set_compute_dtype(previous)   # a module setting: do not leak it into whatever runs next in this process
    t0 = time.perf_counter()
    clusters = list(vc.ClusterGenerator(latent.copy(), lens))
    t_cluster = time.perf_counter() - t0
    q = fd.bin_quality(labels, [c.members for c in clusters], [c.kind_str for c in clusters])
    losses = np.asarray(cap.rows).reshape(-1, 6)
    norms = np.linalg.norm(latent, axis=1)
    q.update(n=n, S=S, nepochs=nepochs, batchsize=bs, batchsteps=steps, dtype=dtype, model_seed=seed, data_seed=dseed,
             t_train=t_train, t_cluster=t_cluster, loss_first=float(losses[0, 0]), loss_last=float(losses[-1, 0]),
             loss_curve=[round(float(x), 6) for x in losses[:, 0]], latent_norm_mean=float(norms.mean()),
             latent_norm_std=float(norms.std()))
    return q


if __name__ == "__main__":
    a = sys.argv[1:]
    q = run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), json.loads(a[4]) or None, a[5], int(a[6]), int(a[7]) if len(a) > 7 else 1)
    line = json.dumps(q)
    print(line)
    if len(a) > 8:
        with open(a[8], "a") as fh:
            fh.write(line + "\n")
