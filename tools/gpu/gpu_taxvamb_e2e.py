"""Diagnostic (GPU box): the whole `vamb bin taxvamb` flow on the product path (vamb/__main__.py:1940-2068) beside `vamb bin default`
on the same synthetic contigs -- loaders -> VAEVAEHLoss.trainmodel -> VAEJoint.encode -> ClusterGenerator -> agreement of the
bins with the synthetic genomes (tests/golden/fixture_defs.bin_quality).  The taxonomy is a random tree whose leaves are the
genomes; a share of the contigs is annotated only to an ancestor of its genome's leaf, or not at all.

    python tools/gpu/gpu_taxvamb_e2e.py N S nepochs batchsize '[batchsteps]' [out.jsonl]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import fixture_defs as fd  # noqa: E402
from vamb_amd import cluster as vc, encode as ve, synth, taxvamb_encode as vt  # noqa: E402


def taxonomy_over_genomes(n_genomes, seed):
    """A tree root -> domain -> ~sqrt(G) phyla -> genera -> one leaf per genome, as ContigTaxonomy-like rank lists."""
    rng = np.random.RandomState(seed)
    n_phyla = max(2, int(round(n_genomes ** 0.5 / 2)))
    ranks = []
    for g in range(n_genomes):
        p = rng.randint(n_phyla)
        genus = rng.randint(4)
        ranks.append(["d_Bacteria", f"p{p}", f"p{p}_g{genus}", f"s{g}"])
    return ranks


class Tax:
    def __init__(self, ranks):
        self.ranks = ranks


def run(n, S, nepochs, bs, steps, seed=1):
    ab, tnf, lens, genome = synth.features(n, S, seed=seed)
    genome = genome.astype(np.int64)
    G = int(genome.max()) + 1
    full = taxonomy_over_genomes(G, seed)
    rng = np.random.RandomState(seed + 7)
    u = rng.random_sample(n)
    depth = np.where(u < 0.15, 0, np.where(u < 0.3, 2, np.where(u < 0.5, 3, 4)))     # unannotated / phylum / genus / species
    taxes = [None if d == 0 else Tax(full[g][:d]) for g, d in zip(genome, depth)]
    nodes, ind, parents = vt.make_graph(taxes)
    targets = np.array([ind["root"] if t is None else ind[t.ranks[-1]] for t in taxes], dtype=np.int64)
    N = len(nodes)
    out = dict(n=n, S=S, genomes=G, taxonomy_nodes=N, nepochs=nepochs, batchsize=bs, batchsteps=steps)
    # --- vamb bin default
    dl_v = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=bs)
    plain = ve.VAE(S, seed=seed)
    t0 = time.perf_counter()
    plain.trainmodel(dl_v, nepochs=nepochs, batchsteps=steps)
    out["default_train_s"] = time.perf_counter() - t0
    lat = plain.encode(dl_v)
    cl = list(vc.ClusterGenerator(lat.copy(), lens))
    q = fd.bin_quality(genome, [c.members for c in cl], [c.kind_str for c in cl])
    out["default"] = {k: q[k] for k in ("n_clusters", "ari", "purity_big", "genomes_recovered", "n_big")}
    # --- vamb bin taxvamb
    dl_j = vt.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, targets, N, parents, batchsize=bs)
    dl_l = vt.make_dataloader_labels_hloss(ab, tnf, lens, targets, N, parents, batchsize=bs)
    dl = vt.make_dataloader_semisupervised_hloss(dl_j, dl_v, dl_l, N, parents, (S, 103, 1, N), seed, batchsize=bs)
    vae = vt.VAEVAEHLoss(S, N, nodes, parents)
    t0 = time.perf_counter()
    vae.trainmodel(dl, nepochs=nepochs, batchsteps=steps)
    out["taxvamb_train_s"] = time.perf_counter() - t0
    out["taxvamb_last_epoch"] = {k: vae.last_epoch_metrics[k] for k in ("loss", "loss_joint", "ce_labels_joint", "loss_vamb", "loss_labels")}
    t0 = time.perf_counter()
    latj = vae.VAEJoint.encode(dl_j)
    clj = list(vc.ClusterGenerator(latj.copy(), lens))
    out["taxvamb_encode_cluster_s"] = time.perf_counter() - t0
    q = fd.bin_quality(genome, [c.members for c in clj], [c.kind_str for c in clj])
    out["taxvamb"] = {k: q[k] for k in ("n_clusters", "ari", "purity_big", "genomes_recovered", "n_big")}
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    res = run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), json.loads(a[4]) or None)
    line = json.dumps(res)
    print(line)
    if len(a) > 5:
        with open(a[5], "a") as fh:
            fh.write(line + "\n")
