#!/usr/bin/env python
"""Records the CLI golden ``tests/golden/cli_bin_default.npz`` + ``cli_bin_default_trace.json`` (build container only).

The reference's REAL command line -- the unmodified ``vamb/__main__.py`` executed by ``oracle/ref_main.py``; ``main()`` ->
``run_bin_default`` -> ``trainvae`` -> ``cluster_and_write_files`` -- runs ``vamb bin default`` on ``.npz`` inputs of
``vamb_amd.synth.features`` with the reference's own classes behind recording proxies (``oracle/cli_reference.py``).  Kept:

* the CALL TRACE: every call the CLI makes on the hot-path names a drop-in rebinds, in the form it makes it (positional /
  keyword, scalar values, array shapes / dtypes) -- what ``vamb_amd``'s classes must accept (``tests/test_cli_dropin.py``
  binds it to their signatures on any machine; ``tests/test_cli_gpu.py`` REPLAYS it on the GPU);
* the latent matrix the reference's VAE wrote (``latent.npz``) and, byte for byte, the three result files its
  ``cluster_and_write_files`` wrote from it -- the expected output of the product's cluster + writer side on the same latent;
* five free-running reference runs (model seeds) of the same command: loss of the last epoch, cluster count and the agreement of
  the bins with the synthetic genomes -- the spread a free-running product run has to land in.

    python tests/golden/make_cli_golden.py            # vamb bin default
    python tests/golden/make_cli_golden.py taxvamb    # vamb bin taxvamb --no_predictor (run_vaevae, vamb/__main__.py:1940-2068):
                                                      # cli_bin_taxvamb.npz + cli_bin_taxvamb_trace.json, same three parts
"""
from __future__ import annotations

import json
import os
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import cli_reference as cr  # noqa: E402
import fixture_defs as fd  # noqa: E402
from vamb_amd import synth  # noqa: E402

CASE = dict(n=4000, nsamples=8, data_seed=5, seed=11, nepochs=12, batchsize=128, batchsteps=[3, 6], threads=4, binsplit="C",
            spread_seeds=[11, 12, 13, 14, 15])
_EPOCH_RE = re.compile(r"Epoch:\s*(\d+)\s+Loss:\s*(\S+)")


def argv_for(case, outdir, comp, abundance, seed):
    return (["bin", "default", "--outdir", str(outdir), "--composition", str(comp), "--abundance", str(abundance),
             "-e", str(case["nepochs"]), "-q"] + [str(q) for q in case["batchsteps"]] +
            ["-t", str(case["batchsize"]), "--seed", str(seed), "-o", case["binsplit"], "-p", str(case["threads"])])


def clusters_of(unsplit_text, names):
    index = {nm: i for i, nm in enumerate(names)}
    by = {}
    for line in unsplit_text.splitlines()[1:]:
        c, m = line.split("\t")
        by.setdefault(c, []).append(index[m])
    return list(by.values())


def main():
    c = CASE
    tmp = tempfile.mkdtemp(prefix="cli_golden_")
    try:
        comp, abundance, names, lens = cr.write_inputs(tmp, c["n"], c["nsamples"], c["data_seed"])
        _, _, _, labels = synth.features(c["n"], c["nsamples"], seed=c["data_seed"])
        spread = []
        first = None
        for seed in c["spread_seeds"]:
            out = os.path.join(tmp, f"out{seed}")
            r = cr.run_cli(argv_for(c, out, comp, abundance, seed), binding="reference")
            files = cr.read_outputs(out)
            losses = [float(m.group(2)) for _, msg in r["log"] for m in [_EPOCH_RE.search(msg)] if m]
            meta = files["vae_clusters_metadata.tsv"].splitlines()[1:]
            kinds = [l.split("\t")[3] for l in meta]
            q = fd.bin_quality(labels, clusters_of(files["vae_clusters_unsplit.tsv"], list(names)), kinds)
            spread.append(dict(seed=seed, loss_last=losses[-1], loss_curve=losses, **{k: q[k] for k in ("n_clusters", "ari", "purity_big", "n_big", "genomes_recovered")}))
            print(spread[-1], flush=True)
            if first is None:
                first = (r, files)
        r, files = first
        trace_path = os.path.join(HERE, "cli_bin_default_trace.json")
        with open(trace_path, "w") as fh:
            json.dump(dict(case=c, argv=argv_for(c, "<outdir>", "<composition.npz>", "<abundance.npz>", c["seed"]),
                           trace=r["trace"], files=files["files"],
                           log=[m for _, m in r["log"] if "seconds" not in m and "Invoked with" not in m]), fh, indent=1)
        np.savez_compressed(
            os.path.join(HERE, "cli_bin_default.npz"),
            latent=files["latent"], names=np.array(list(names), dtype="U"), lengths=lens,
            metadata_tsv=np.array(files["vae_clusters_metadata.tsv"]), unsplit_tsv=np.array(files["vae_clusters_unsplit.tsv"]),
            split_tsv=np.array(files["vae_clusters_split.tsv"]),
            spread_seed=np.array([s["seed"] for s in spread]), spread_loss_last=np.array([s["loss_last"] for s in spread]),
            spread_loss_curve=np.array([s["loss_curve"] for s in spread]),
            spread_n_clusters=np.array([s["n_clusters"] for s in spread]), spread_ari=np.array([s["ari"] for s in spread]),
            spread_purity_big=np.array([s["purity_big"] for s in spread]), spread_n_big=np.array([s["n_big"] for s in spread]),
            spread_genomes_recovered=np.array([s["genomes_recovered"] for s in spread]))
        print("wrote", trace_path, "and cli_bin_default.npz;", len(r["trace"]), "calls recorded")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


CASE_TAX = dict(n=3000, nsamples=6, data_seed=7, seed=21, nepochs=8, batchsize=128, batchsteps=[2, 5], threads=4, binsplit="C",
                spread_seeds=[21, 22, 23])
_TAX_LOSS_RE = re.compile(r"Epoch:\s*(\d+)\s.*\sloss: (\S+)\s*$")


def argv_for_taxvamb(case, outdir, comp, abundance, taxonomy, seed):
    return (["bin", "taxvamb", "--outdir", str(outdir), "--composition", str(comp), "--abundance", str(abundance),
             "--taxonomy", str(taxonomy), "--no_predictor", "-e", str(case["nepochs"]), "-q"] + [str(q) for q in case["batchsteps"]] +
            ["-t", str(case["batchsize"]), "--seed", str(seed), "-o", case["binsplit"], "-p", str(case["threads"])])


def main_taxvamb():
    c = CASE_TAX
    tmp = tempfile.mkdtemp(prefix="cli_golden_tax_")
    try:
        comp, abundance, names, lens = cr.write_inputs(tmp, c["n"], c["nsamples"], c["data_seed"])
        _, _, _, labels = synth.features(c["n"], c["nsamples"], seed=c["data_seed"])
        taxonomy = cr.write_taxonomy(tmp, names, labels)
        spread = []
        first = None
        for seed in c["spread_seeds"]:
            out = os.path.join(tmp, f"out{seed}")
            r = cr.run_cli(argv_for_taxvamb(c, out, comp, abundance, taxonomy, seed), binding="reference", family="taxvamb")
            files = cr.read_outputs(out, prefix="vaevae", latent_name="vaevae_latent.npz")
            losses = [float(m.group(2)) for _, msg in r["log"] for m in [_TAX_LOSS_RE.search(msg)] if m]
            meta = files["vaevae_clusters_metadata.tsv"].splitlines()[1:]
            kinds = [l.split("\t")[3] for l in meta]
            q = fd.bin_quality(labels, clusters_of(files["vaevae_clusters_unsplit.tsv"], list(names)), kinds)
            spread.append(dict(seed=seed, loss_last=losses[-1], loss_curve=losses, **{k: q[k] for k in ("n_clusters", "ari", "purity_big", "n_big", "genomes_recovered")}))
            print(spread[-1], flush=True)
            if first is None:
                first = (r, files)
        r, files = first
        trace_path = os.path.join(HERE, "cli_bin_taxvamb_trace.json")
        with open(trace_path, "w") as fh:
            json.dump(dict(case=c, argv=argv_for_taxvamb(c, "<outdir>", "<composition.npz>", "<abundance.npz>", "<taxonomy.tsv>", c["seed"]),
                           trace=r["trace"], files=files["files"],
                           log=[m for _, m in r["log"] if "seconds" not in m and "Invoked with" not in m]), fh, indent=1)
        np.savez_compressed(
            os.path.join(HERE, "cli_bin_taxvamb.npz"),
            latent=files["latent"], names=np.array(list(names), dtype="U"), lengths=lens,
            taxonomy_lines=np.array(cr.taxonomy_lines(names, labels), dtype="U"),
            metadata_tsv=np.array(files["vaevae_clusters_metadata.tsv"]), unsplit_tsv=np.array(files["vaevae_clusters_unsplit.tsv"]),
            split_tsv=np.array(files["vaevae_clusters_split.tsv"]),
            spread_seed=np.array([s["seed"] for s in spread]), spread_loss_last=np.array([s["loss_last"] for s in spread]),
            spread_loss_curve=np.array([s["loss_curve"] for s in spread]),
            spread_n_clusters=np.array([s["n_clusters"] for s in spread]), spread_ari=np.array([s["ari"] for s in spread]),
            spread_purity_big=np.array([s["purity_big"] for s in spread]), spread_n_big=np.array([s["n_big"] for s in spread]),
            spread_genomes_recovered=np.array([s["genomes_recovered"] for s in spread]))
        print("wrote", trace_path, "and cli_bin_taxvamb.npz;", len(r["trace"]), "calls recorded")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "taxvamb":
        main_taxvamb()
    else:
        main()
