#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (oracle/): calibrate bench.py's CPU port against the REAL reference.

bench.py's `cpu_baseline` leg times the oracle restatements ("port": oracle/vae_oracle.py in numpy fp32 +
oracle/cluster_scan.c) because /root/reference does not exist on the GPU box.  This script, run in the BUILD
container where the reference does exist, runs the reference itself (oracle/ref_harness.py: the unmodified
vamb/{encode,cluster}.py with torch on the CPU, `cuda=False`) and the port on the SAME 20 k-contig sample with
the same thread count and writes the ratio to oracle/cpu_calibration.json.  bench.py reports it as
`cpu_baseline.calibration_vs_reference` (reference time / port time, per stage and for the whole job).

    python oracle/calibrate_cpu_baseline.py [--contigs 20000,7000] [--samples 200] [--batch 8192] [--threads 8]

Several sample sizes (round 5): the top-level fields of the JSON describe the FIRST size; `points` holds every size, and
`cluster_time_exponent` the exponent p of t = c n^p fitted through the sizes for the reference's sweep and for the port's
(the sweep is the one stage whose cost is not linear in the number of contigs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--contigs", type=str, default="20000,7000", help="comma-separated sample sizes (the first one is the headline)")
    p.add_argument("--samples", type=int, default=200)
    p.add_argument("--batch", type=int, default=8192)
    p.add_argument("--latent", type=int, default=32)
    p.add_argument("--epochs", type=int, default=3)
    p.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    args = p.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(args.threads))
    os.environ.setdefault("MKL_NUM_THREADS", str(args.threads))
    import torch

    torch.set_num_threads(args.threads)
    import cluster_oracle as co
    import ref_harness
    import vae_oracle as vo
    from vamb_amd import synth

    vt, rc, re_ = ref_harness.load_reference()
    sizes = [int(x) for x in args.contigs.split(",")]
    points = [calibrate_one(args, n, torch, co, ref_harness, vo, synth, rc, re_) for n in sizes]
    out = dict(points[0])
    out["points"] = points
    if len(points) > 1:
        ln = np.log([float(q["contigs"]) for q in points])
        out["cluster_time_exponent"] = {
            k: float(np.polyfit(ln, np.log([q[k]["cluster_s"] for q in points]), 1)[0]) for k in ("reference", "port")}
        out["cluster_time_exponent"]["sizes"] = sizes
    with open(os.path.join(HERE, "cpu_calibration.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def calibrate_one(args, n, torch, co, ref_harness, vo, synth, rc, re_):
    S, bs = args.samples, min(args.batch, n)
    ab, tnf, lens, _ = synth.features(n, S, seed=101)

    # ---- the reference itself: make_dataloader -> VAE.trainmodel -> encode -> list(ClusterGenerator)
    t0 = time.perf_counter()
    dl = re_.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=bs, destroy=True, cuda=False)
    t_prep = time.perf_counter() - t0
    vae = re_.VAE(S, nlatent=args.latent, cuda=False, seed=0)
    t0 = time.perf_counter()
    vae.trainmodel(dl, nepochs=args.epochs, batchsteps=None)
    ref_epoch = (time.perf_counter() - t0) / args.epochs
    t0 = time.perf_counter()
    latent = vae.encode(dl)
    ref_encode = time.perf_counter() - t0
    # cluster a structured latent (blob latents of the same size: the 3-epoch model's latents are unstructured)
    lat, _ = synth.blob_latent(n, args.latent, sigma=0.08, seed=5)
    t0 = time.perf_counter()
    ref_clusters = sum(1 for _ in rc.ClusterGenerator(lat.copy(), lens, destroy=True, rng_seed=1))
    ref_cluster = time.perf_counter() - t0

    # ---- the port, same sample, same threads
    from vamb_amd import encode as ve_host  # make_dataloader is host numpy (no GPU needed)

    dl2 = ve_host.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=bs, destroy=True)
    d, t, a, w = (x.numpy() for x in dl2.dataset.tensors)
    st = vo.init_state(S, [512, 512], args.latent, 1)
    m = vo.OracleVAE(S, [512, 512], args.latent, None, 200.0, 0.2, state=st, dtype=np.float32)
    rng = np.random.RandomState(0)
    nb = max(1, n // bs)
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        perm = rng.permutation(n)
        for b in range(nb):
            rows = perm[b * bs:(b + 1) * bs]
            masks = [rng.random_sample((len(rows), 512)) >= 0.2 for _ in range(4)]
            eps = rng.standard_normal((len(rows), args.latent)).astype(np.float32)
            m.train_step(d[rows], t[rows], a[rows], w[rows], eps, masks)
    port_epoch = (time.perf_counter() - t0) / args.epochs
    t0 = time.perf_counter()
    m.encode(d, t, a)
    port_encode = time.perf_counter() - t0
    t0 = time.perf_counter()
    port_clusters = sum(1 for _ in co.OracleClusterGenerator(lat.copy(), lens, rng_seed=1))
    port_cluster = time.perf_counter() - t0
    assert port_clusters == ref_clusters, (port_clusters, ref_clusters)

    E = 300
    ref_total = E * ref_epoch + ref_encode + ref_cluster
    port_total = E * port_epoch + port_encode + port_cluster
    return dict(
        contigs=n,
        sample=f"{n} contigs x {S} samples, batch {bs}, latent {args.latent}, {args.epochs} epochs timed, {ref_clusters} clusters "
               f"(blob latents sigma 0.08)",
        threads=args.threads, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
        torch=torch.__version__, numpy=np.__version__,
        reference=dict(make_dataloader_s=t_prep, epoch_s=ref_epoch, encode_s=ref_encode, cluster_s=ref_cluster,
                       job_300_epochs_s=ref_total),
        port=dict(epoch_s=port_epoch, encode_s=port_encode, cluster_s=port_cluster, job_300_epochs_s=port_total),
        reference_over_port=dict(epoch=ref_epoch / port_epoch, encode=ref_encode / port_encode,
                                 cluster=ref_cluster / port_cluster, job_300_epochs=ref_total / port_total),
    )


if __name__ == "__main__":
    main()
