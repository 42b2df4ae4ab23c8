"""TEST INFRASTRUCTURE ONLY -- end-to-end runs of the REAL reference.

``make_dataloader -> VAE.trainmodel -> VAE.encode -> list(ClusterGenerator)`` through ``oracle/ref_harness.py`` on the
synthetic features of ``vamb_amd/synth.py``, free-running RNG.  Records what a statistical comparison needs: the loss
curve (parsed from the reference's own epoch log line, vamb/encode.py:427-437), the cluster count by kind and the
agreement of the bins with the synthetic genomes (adjusted Rand index, purity).  Build container only.

    python oracle/e2e_reference.py N S nepochs batchsize '[batchsteps]' model_seed [data_seed] [threads]   # one run, JSON line
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

_EPOCH_RE = re.compile(r"Epoch:\s*(\d+)\s+Loss:\s*(\S+)\s+CE:\s*(\S+)\s+AB:\s*(\S+)\s+SSE:\s*(\S+)\s+KLD:\s*(\S+)\s+Batchsize:\s*(\d+)")


class EpochLog:
    """Drop-in for the module-level ``logger`` of vamb/encode.py (or a logging.Handler target): keeps the epoch lines."""

    def __init__(self):
        self.rows = []

    def info(self, msg, *a, **k):
        m = _EPOCH_RE.search(str(msg))
        if m:
            self.rows.append([float(m.group(i)) for i in (2, 3, 4, 5, 6)] + [float(m.group(7))])
        return self

    def __getattr__(self, name):   # debug / warning / opt / ... : no-ops
        return lambda *a, **k: self

    def array(self) -> np.ndarray:
        """[epochs][6]: loss, ce, ab, sse, kld, batchsize"""
        return np.asarray(self.rows, dtype=np.float64).reshape(-1, 6)


def run_reference(n, nsamples, nepochs, batchsize, batchsteps, model_seed, data_seed=1, threads=8):
    import torch

    import fixture_defs as fd
    import ref_harness
    from vamb_amd import synth

    _, cl, en = ref_harness.load_reference()
    torch.set_num_threads(threads)
    ab, tnf, lens, labels = synth.features(n, nsamples, seed=data_seed)
    log = EpochLog()
    saved = en.logger
    en.logger = log
    try:
        dl = en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=batchsize)
        vae = en.VAE(nsamples, seed=model_seed)
        t0 = time.perf_counter()
        vae.trainmodel(dl, nepochs=nepochs, batchsteps=batchsteps)
        t_train = time.perf_counter() - t0
        latent = vae.encode(dl)
    finally:
        en.logger = saved
    t0 = time.perf_counter()
    clusters = list(cl.ClusterGenerator(latent.copy(), lens))
    t_cluster = time.perf_counter() - t0
    q = fd.bin_quality(labels, [c.members for c in clusters], [c.kind_str for c in clusters])
    q.update(t_train=t_train, t_cluster=t_cluster, losses=log.array())
    return q


if __name__ == "__main__":
    a = sys.argv[1:]
    n, S, nep, bs = int(a[0]), int(a[1]), int(a[2]), int(a[3])
    steps = json.loads(a[4])
    seed = int(a[5])
    dseed = int(a[6]) if len(a) > 6 else 1
    thr = int(a[7]) if len(a) > 7 else 8
    q = run_reference(n, S, nep, bs, steps, seed, dseed, thr)
    losses = q.pop("losses")
    q.update(n=n, S=S, nepochs=nep, batchsize=bs, batchsteps=steps, model_seed=seed, data_seed=dseed,
             loss_first=losses[0, 0], loss_last=losses[-1, 0], loss_curve=[round(float(x), 6) for x in losses[:, 0]])
    print(json.dumps(q))
