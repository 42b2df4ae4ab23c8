"""``vamb bin default`` REPLAYED on the GPU (the other half of ``tests/test_cli_dropin.py``).

The reference tree cannot travel to the GPU box, so its function bodies cannot run here.  What travels is DATA recorded in the
build container from the reference's real ``main()`` (``tests/golden/make_cli_golden.py``): the call trace of ``run_bin_default`` /
``trainvae`` / ``cluster_and_write_files`` on the hot-path names (``vamb/__main__.py:1458, 1075, 1088-1097, 1277``), the latent the
reference's VAE wrote, the three result files the reference's writer produced from it, and five free-running reference runs.
Here the trace is replayed call by call, in the recorded positional / keyword form, on the product's classes and the device
library: (a) every call is accepted and returns what the CLI's next statement uses; (b) on the SAME latent the product's
generator + writer produce the reference's files byte for byte; (c) the free-running run lands in the reference's spread."""
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fixture_defs as fd  # noqa: E402

pytestmark = pytest.mark.gpu
TRACE = json.load(open(os.path.join(HERE, "golden", "cli_bin_default_trace.json")))


class _BinSplitter:
    """The interface of ``vamb.vambtools.BinSplitter`` the writer uses (vambtools.py:27-141), initialised with separator "C"."""

    def __init__(self, splitter):
        self.splitter = splitter
        self.logged = None

    def is_disabled(self):
        return self.splitter is None

    def log_string(self):
        return "None" if self.splitter is None else f'"{self.splitter}"'

    def log_clustering_result(self, n_total, n_split, n_unsplit, begintime):
        self.logged = (n_total, n_split, n_unsplit)


def _materialise(desc, pool):
    """A recorded argument -> a live object: scalars by value, arrays / loaders from ``pool`` by kind."""
    if isinstance(desc, dict):
        return pool[desc["kind"]]()
    return desc


def _clusters_of(unsplit_text, names):
    index = {nm: i for i, nm in enumerate(names)}
    by = {}
    for line in unsplit_text.splitlines()[1:]:
        c, m = line.split("\t")
        by.setdefault(c, []).append(index[m])
    return list(by.values())


def test_replay_of_the_real_clis_calls(tmp_path):
    from vamb_amd import cluster as vc, encode as ve, output, synth

    c = TRACE["case"]
    g = np.load(os.path.join(HERE, "golden", "cli_bin_default.npz"))
    ab, tnf, lens, labels = synth.features(c["n"], c["nsamples"], seed=c["data_seed"])
    lens32 = lens.astype(np.int32)                      # CompositionMetaData.lengths as Composition.load returns it
    assert np.array_equal(lens, g["lengths"])
    names = [str(x) for x in g["names"]]
    state = {}
    arrays = iter([ab, tnf, lens32])                    # make_dataloader(abundance.matrix, composition.matrix, lengths, ...)
    pool = {"ndarray": lambda: next(arrays), "DataLoader": lambda: state["loader"], "Path": lambda: Path(tmp_path) / "model.pt"}
    fns = {"vamb.encode.make_dataloader": ve.make_dataloader, "vamb.encode.set_batchsize": ve.set_batchsize,
           "vamb.encode.VAE": ve.VAE}
    calls = [x for x in TRACE["trace"]]
    # the CLI's own calls: set_batchsize calls made from INSIDE the reference's trainmodel / encode are the product's business
    cli_calls = calls[:4] + [x for x in calls[4:] if x["name"] != "vamb.encode.set_batchsize"]
    latent = None
    for call in cli_calls:
        name = call["name"]
        if name == "vamb.cluster.ClusterGenerator":   # (its two arrays are the latent and the lengths: bound in part (b) below)
            assert [a["kind"] for a in call["args"]] == ["ndarray", "ndarray"]
            gen_kwargs = {k: _materialise(v, pool) for k, v in call["kwargs"].items()}
            continue
        args = [_materialise(a, pool) for a in call["args"]]
        kwargs = {k: _materialise(v, pool) for k, v in call["kwargs"].items()}
        if name == "vamb.encode.make_dataloader":
            state["loader"] = fns[name](*args, **kwargs)
            got = [(list(t.shape), str(t.dtype)) for t in state["loader"].dataset.tensors]
            assert got == [(t["shape"], t["dtype"]) for t in call["result"]["tensors"]]      # __main__.py:1073-1074 reads these
            assert state["loader"].batch_size == call["result"]["batch_size"]
        elif name == "vamb.encode.VAE":
            state["vae"] = fns[name](*args, **kwargs)
        elif name == "vamb.encode.set_batchsize":
            state["train_loader"] = fns[name](*args, **kwargs)
            assert state["train_loader"].batch_size == call["result"]["batch_size"]
        elif name == "VAE.trainmodel":
            assert args[0] is state["loader"]            # (recorded as "a DataLoader": the CLI passes set_batchsize's result)
            state["vae"].trainmodel(state["train_loader"], **kwargs)
            assert (Path(tmp_path) / "model.pt").is_file()
        elif name == "VAE.encode":
            latent = state["vae"].encode(*args, **kwargs)
            assert isinstance(latent, np.ndarray) and list(latent.shape) == call["result"]["shape"]
            assert str(latent.dtype) == call["result"]["dtype"] and np.isfinite(latent).all()
        else:
            raise AssertionError(name)
    # the model file reloads into the product's class (trainvae:1093 leaves it for `vamb recluster` / later runs)
    again = ve.VAE.load(str(Path(tmp_path) / "model.pt"))
    assert again.nsamples == c["nsamples"] and np.array_equal(again.encode(state["loader"]), latent)

    # (b) cluster_and_write_files (vamb/__main__.py:1254-1404) on the latent the REFERENCE wrote: byte-identical result files
    opts = types.SimpleNamespace(window_size=gen_kwargs["windowsize"], min_successes=gen_kwargs["minsuccesses"], max_clusters=None)
    splitter = _BinSplitter(c["binsplit"])
    base = str(Path(tmp_path) / "vae_clusters")
    output.cluster_and_write_files(opts, splitter, g["latent"].copy(), names, lens32, gen_kwargs["rng_seed"], gen_kwargs["cuda"],
                                   base, None, None)
    assert open(base + "_metadata.tsv").read() == str(g["metadata_tsv"])
    assert open(base + "_unsplit.tsv").read() == str(g["unsplit_tsv"])
    a, b = open(base + "_split.tsv").read().splitlines(), str(g["split_tsv"]).splitlines()
    assert a[0] == b[0] and sorted(a) == sorted(b)
    assert list(dict.fromkeys(l.split("\t")[0] for l in a)) == list(dict.fromkeys(l.split("\t")[0] for l in b))
    # the generator itself with the CLI's exact keyword form
    stream = list(vc.ClusterGenerator(g["latent"].copy(), lens32, **gen_kwargs))
    assert len(stream) == len(str(g["metadata_tsv"]).splitlines()) - 1

    # (c) free-running: the product's latent through the product's writer, against five runs of the reference's CLI
    base2 = str(Path(tmp_path) / "free")
    output.cluster_and_write_files(opts, _BinSplitter(c["binsplit"]), latent.copy(), names, lens32, gen_kwargs["rng_seed"], False,
                                   base2, None, None)
    meta = open(base2 + "_metadata.tsv").read().splitlines()[1:]
    q = fd.bin_quality(labels, _clusters_of(open(base2 + "_unsplit.tsv").read(), names), [l.split("\t")[3] for l in meta])
    last = state["vae"].last_epoch_losses["loss"]
    ref_last = g["spread_loss_last"]
    assert abs(last - ref_last.mean()) < max(3 * (ref_last.max() - ref_last.min()), 5e-3), (last, ref_last)
    assert q["ari"] >= g["spread_ari"].min() - 0.03, (q["ari"], g["spread_ari"])
    assert q["purity_big"] >= g["spread_purity_big"].min() - 0.01
    assert 0.7 * g["spread_n_clusters"].min() <= q["n_clusters"] <= 1.3 * g["spread_n_clusters"].max(), q["n_clusters"]
    assert q["genomes_recovered"] >= g["spread_genomes_recovered"].min() - 3


# ---- `vamb bin taxvamb --no_predictor` (run_vaevae, vamb/__main__.py:1940-2068) replayed with install(semisupervised=True)'s classes ----
TRACE_TAX = json.load(open(os.path.join(HERE, "golden", "cli_bin_taxvamb_trace.json")))


def test_replay_of_the_real_taxvamb_clis_calls(tmp_path):
    """The recorded calls of the reference's run_vaevae, in their positional / keyword form, on the product's TaxVamb classes:
    (a) every call is accepted and returns what the CLI's next statement uses (loader tensors, the joint latent's shape);
    (b) on the latent the REFERENCE wrote the product's generator + writer give the reference's files byte for byte;
    (c) the free-running joint trainer's last loss and the bins of its latent land in the spread of three reference CLI runs."""
    from vamb_amd import encode as ve, output, synth, taxvamb_encode as vt

    c = TRACE_TAX["case"]
    g = np.load(os.path.join(HERE, "golden", "cli_bin_taxvamb.npz"))
    ab, tnf, lens, labels = synth.features(c["n"], c["nsamples"], seed=c["data_seed"])
    lens32 = lens.astype(np.int32)
    assert np.array_equal(lens, g["lengths"])
    names = [str(x) for x in g["names"]]
    # Taxonomy.from_file (vamb/taxonomy.py:61-120, 30-35): one ContigTaxonomy per contig, ranks split at ';', [] for an empty field
    taxes = []
    for nm, line in zip(names, (str(x) for x in g["taxonomy_lines"])):
        contig, pred = line.split("\t")
        assert contig == nm
        taxes.append(types.SimpleNamespace(ranks=pred.split(";") if pred else []))
    nodes, ind_nodes, table_parent = vt.make_graph(taxes)                      # __main__.py:1981-1983
    targets = np.array([ind_nodes["root" if len(t.ranks) == 0 else t.ranks[-1]] for t in taxes])   # :1984-1990

    calls = TRACE_TAX["trace"]
    # (the two bare make_dataloader calls come from INSIDE the reference's concat / labels loaders: the product's business)
    cli_calls = [x for x in calls if not (x["name"] == "vamb.encode.make_dataloader" and not x["kwargs"])]
    state = {}

    def live(desc, arrays, lists, loaders):
        if isinstance(desc, dict):
            kind = desc["kind"]
            if kind == "ndarray":
                return next(arrays)
            if kind == "list":
                return lists[desc["of"]]
            if kind == "DataLoader":
                return next(loaders)
            if kind == "BufferedWriter":
                return state["modelfile"]
            raise AssertionError(kind)
        return desc

    def tensors_of(loader):
        return [(list(t.shape), str(t.dtype)) for t in loader.dataset.tensors]

    latent = None
    gen_kwargs = None
    for call in cli_calls:
        name = call["name"]
        arrays = iter([ab, tnf, lens32, targets])
        lists = {"str": nodes, "int": table_parent}
        loaders = iter([state.get("joint"), state.get("vamb"), state.get("labels")])
        if name == "vamb.cluster.ClusterGenerator":
            assert [a["kind"] for a in call["args"]] == ["ndarray", "ndarray"]
            gen_kwargs = dict(call["kwargs"])
            continue
        if name == "VAEVAEHLoss.trainmodel":
            loaders = iter([state["semi"]])
            state["modelfile"] = open(Path(tmp_path) / "vaevae_model.pt", "wb")
        if name == "VAEJoint.encode":
            loaders = iter([state["joint"]])
        args = [live(a, arrays, lists, loaders) for a in call["args"]]
        kwargs = {k: live(v, arrays, lists, loaders) for k, v in call["kwargs"].items()}
        if name == "vamb.taxvamb_encode.VAEVAEHLoss":
            assert args[1] == len(nodes) and call["args"][2]["len"] == len(nodes) and call["args"][3]["len"] == len(table_parent)
            state["vae"] = vt.VAEVAEHLoss(*args, **kwargs)
        elif name == "vamb.encode.make_dataloader":
            state["vamb"] = ve.make_dataloader(*args, **kwargs)
            assert tensors_of(state["vamb"]) == [(t["shape"], t["dtype"]) for t in call["result"]["tensors"]]
        elif name == "vamb.taxvamb_encode.make_dataloader_concat_hloss":
            state["joint"] = vt.make_dataloader_concat_hloss(*args, **kwargs)
            assert tensors_of(state["joint"]) == [(t["shape"], t["dtype"]) for t in call["result"]["tensors"]]
        elif name == "vamb.taxvamb_encode.make_dataloader_labels_hloss":
            state["labels"] = vt.make_dataloader_labels_hloss(*args, **kwargs)
            assert tensors_of(state["labels"]) == [(t["shape"], t["dtype"]) for t in call["result"]["tensors"]]
        elif name == "vamb.taxvamb_encode.make_dataloader_semisupervised_hloss":
            assert args[0] is state["joint"] and args[1] is state["vamb"] and args[2] is state["labels"]
            state["semi"] = vt.make_dataloader_semisupervised_hloss(*args, **kwargs)
            assert state["semi"].batch_size == call["result"]["batch_size"]
        elif name == "VAEVAEHLoss.trainmodel":
            state["vae"].trainmodel(*args, **kwargs)
            state["modelfile"].close()
            assert (Path(tmp_path) / "vaevae_model.pt").stat().st_size > 0
        elif name == "VAEJoint.encode":
            latent = state["vae"].VAEJoint.encode(*args, **kwargs)
            assert isinstance(latent, np.ndarray) and list(latent.shape) == call["result"]["shape"]
            assert str(latent.dtype) == call["result"]["dtype"] and np.isfinite(latent).all()
        else:
            raise AssertionError(name)
    assert latent is not None and gen_kwargs is not None

    # (b) cluster_and_write_files (vamb/__main__.py:2055-2066 -> 1254-1404) on the latent the REFERENCE wrote
    opts = types.SimpleNamespace(window_size=gen_kwargs["windowsize"], min_successes=gen_kwargs["minsuccesses"], max_clusters=None)
    base = str(Path(tmp_path) / "vaevae_clusters")
    output.cluster_and_write_files(opts, _BinSplitter(c["binsplit"]), g["latent"].copy(), names, lens32, gen_kwargs["rng_seed"],
                                   gen_kwargs["cuda"], base, None, None)
    assert open(base + "_metadata.tsv").read() == str(g["metadata_tsv"])
    assert open(base + "_unsplit.tsv").read() == str(g["unsplit_tsv"])
    a, b = open(base + "_split.tsv").read().splitlines(), str(g["split_tsv"]).splitlines()
    assert a[0] == b[0] and sorted(a) == sorted(b)

    # (c) free-running
    base2 = str(Path(tmp_path) / "free")
    output.cluster_and_write_files(opts, _BinSplitter(c["binsplit"]), latent.copy(), names, lens32, gen_kwargs["rng_seed"], False,
                                   base2, None, None)
    meta = open(base2 + "_metadata.tsv").read().splitlines()[1:]
    q = fd.bin_quality(labels, _clusters_of(open(base2 + "_unsplit.tsv").read(), names), [l.split("\t")[3] for l in meta])
    last = state["vae"].last_epoch_metrics["loss"]
    ref_last = g["spread_loss_last"]
    assert abs(last - ref_last.mean()) < max(3 * (ref_last.max() - ref_last.min()), 2e-2), (last, ref_last)
    assert q["ari"] >= g["spread_ari"].min() - 0.05, (q["ari"], g["spread_ari"])
    assert q["purity_big"] >= g["spread_purity_big"].min() - 0.02
    assert 0.6 * g["spread_n_clusters"].min() <= q["n_clusters"] <= 1.5 * g["spread_n_clusters"].max(), q["n_clusters"]
    assert q["genomes_recovered"] >= g["spread_genomes_recovered"].min() - 3
