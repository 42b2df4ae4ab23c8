"""The reference's REAL command line against the drop-in (``vamb/__main__.py:1065-1107, 1254-1404, 1451-1488``).

Two machines, two halves -- ``/root/reference`` exists only in the build container (no GPU), the GPU box has no reference tree:

* HERE (build container): ``oracle/ref_main.py`` imports the UNMODIFIED ``vamb/__main__.py`` (stubs for pycoverm / pyhmmer /
  pyrodigal, which ``vamb bin default`` on .npz inputs never calls) and ``oracle/cli_reference.py`` runs its ``main()``.
  ``tests/golden/make_cli_golden.py`` recorded, with the reference's own classes bound, the CALL TRACE of the CLI on the hot-path
  names, the latent it wrote and the result files.  The tests below (i) bind every recorded call to the product's signatures,
  (ii) check that ``dropin.install`` rebinds the names of the really imported ``vamb.__main__``, and (iii) run the real ``main()``
  once more with ``dropin.install()`` active: the product's ``ClusterGenerator`` (its host logic; scans answered by the oracle
  backend, there being no GPU here) and the product's ``cluster_and_write_files`` inside the reference's ``run_bin_default`` --
  same files, byte for byte.
* ON THE GPU BOX: ``tests/test_cli_gpu.py`` replays the same call trace on the product's classes with the device library.
"""
import inspect
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_harness  # noqa: E402

TRACE = json.load(open(os.path.join(HERE, "golden", "cli_bin_default_trace.json")))
needs_reference = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


def _targets():
    from vamb_amd import cluster as vc, encode as ve

    return {"vamb.encode.make_dataloader": (ve.make_dataloader, False), "vamb.encode.set_batchsize": (ve.set_batchsize, False),
            "vamb.encode.VAE": (ve.VAE.__init__, True), "VAE.trainmodel": (ve.VAE.trainmodel, True),
            "VAE.encode": (ve.VAE.encode, True), "vamb.cluster.ClusterGenerator": (vc.ClusterGenerator.__init__, True)}


def test_every_recorded_cli_call_binds_to_the_products_signatures():
    """Runs anywhere: the trace is data.  Every call of the real CLI (and of the reference's trainmodel / encode on
    set_batchsize) must be a valid call of the product's function of the same name: same positional order, same keywords."""
    targets = _targets()
    seen = set()
    for call in TRACE["trace"]:
        fn, is_method = targets[call["name"]]
        args = ([object()] if is_method else []) + list(call["args"])
        bound = inspect.signature(fn).bind(*args, **call["kwargs"])   # TypeError = the drop-in would reject the CLI's call
        assert not any(k.startswith("_") for k in bound.arguments), "the CLI never passes a private parameter"
        seen.add(call["name"])
    assert seen == set(targets), "the recorded run exercised every hot-path name"
    # the CLI's own call order (vamb/__main__.py:1458, 1075, 1089, 1088, 1096, 1277)
    outer = [c["name"] for c in TRACE["trace"] if not (c["name"] == "vamb.encode.set_batchsize" and c is not TRACE["trace"][2])]
    assert outer == ["vamb.encode.make_dataloader", "vamb.encode.VAE", "vamb.encode.set_batchsize", "VAE.trainmodel", "VAE.encode",
                     "vamb.cluster.ClusterGenerator"]


@needs_reference
def test_install_rebinds_the_really_imported_cli_module():
    import ref_main
    from vamb_amd import cluster as vc, dropin, encode as ve, output

    vamb, main = ref_main.load_reference_main()
    assert main.__file__.endswith(os.path.join("vamb", "__main__.py")) and callable(main.run_bin_default)
    ref_writer, ref_vae, ref_cg = main.cluster_and_write_files, vamb.encode.VAE, vamb.cluster.ClusterGenerator
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")          # nothing may be left on the reference's path silently
        saved = dropin.install(vamb, semisupervised=True, strict=True)
    try:
        assert main.cluster_and_write_files is output.cluster_and_write_files
        assert vamb.encode.VAE is ve.VAE and vamb.cluster.ClusterGenerator is vc.ClusterGenerator
        # the names the CLI's function bodies look up at call time resolve to the product's objects
        assert main.vamb.encode.make_dataloader is ve.make_dataloader and main.vamb.encode.set_batchsize is ve.set_batchsize
        from vamb_amd import semisupervised_encode as vs, taxvamb_encode as vt

        assert vamb.taxvamb_encode.VAEVAEHLoss is vt.VAEVAEHLoss and vamb.semisupervised_encode.VAEVAE is vs.VAEVAE
        # same parameters as the reference's writer (names and defaults; ours adds private hooks only)
        ours = [(p.name, p.default) for p in inspect.signature(output.cluster_and_write_files).parameters.values()
                if not p.name.startswith("_")]
        theirs = [(p.name, p.default) for p in inspect.signature(ref_writer).parameters.values()]
        assert ours == theirs
    finally:
        dropin.uninstall(saved, vamb)
    assert main.cluster_and_write_files is ref_writer and vamb.encode.VAE is ref_vae and vamb.cluster.ClusterGenerator is ref_cg


@needs_reference
def test_real_cli_with_the_drop_in_bound_writes_the_reference_files(tmp_path):
    """``vamb bin default`` -- the reference's main(), option classes, run_bin_default, trainvae -- with dropin.install() active.
    No GPU here: the VAE stays the reference's (same seed, same threads: the same latent as the golden run, checked), the
    product's ClusterGenerator runs on the oracle backend, the product's writer replaces cluster_and_write_files."""
    import cli_reference as cr
    import make_cli_golden as mk
    from oracle_backend import OracleScanBackend

    g = np.load(os.path.join(HERE, "golden", "cli_bin_default.npz"))
    c = TRACE["case"]
    comp, abundance, names, _ = cr.write_inputs(str(tmp_path), c["n"], c["nsamples"], c["data_seed"])
    assert list(names) == list(g["names"])
    out = tmp_path / "out"
    r = cr.run_cli(mk.argv_for(c, out, comp, abundance, c["seed"]), binding="dropin", vae="reference",
                   backend_factory=OracleScanBackend)
    files = cr.read_outputs(out)
    assert files["files"] == TRACE["files"]
    if not np.array_equal(files["latent"], g["latent"]):
        pytest.skip("the reference's CPU training did not reproduce the golden latent bit for bit on this host")
    assert files["vae_clusters_metadata.tsv"] == str(g["metadata_tsv"])
    assert files["vae_clusters_unsplit.tsv"] == str(g["unsplit_tsv"])
    a, b = files["vae_clusters_split.tsv"].splitlines(), str(g["split_tsv"]).splitlines()
    assert a[0] == b[0] and sorted(a) == sorted(b)   # (a set per split bin in the reference: hash-seed dependent line order)
    assert list(dict.fromkeys(l.split("\t")[0] for l in a)) == list(dict.fromkeys(l.split("\t")[0] for l in b))
    # what the user sees: the log of the run, timing lines apart
    def visible(lines):   # (wall-clock lines, the temporary directory's paths and the epoch losses apart)
        return [m for m in lines if "seconds" not in m and "Invoked with" not in m and ".npz" not in m and "Epoch:" not in m]

    assert visible(m for _, m in r["log"]) == visible(TRACE["log"])


# ---- `vamb bin taxvamb --no_predictor`: run_vaevae (vamb/__main__.py:1940-2068) with install(semisupervised=True) ----------------
TRACE_TAX = json.load(open(os.path.join(HERE, "golden", "cli_bin_taxvamb_trace.json")))


def _targets_taxvamb():
    from vamb_amd import cluster as vc, encode as ve, semisupervised_encode as vs, taxvamb_encode as vt

    return {"vamb.encode.make_dataloader": (ve.make_dataloader, False),
            "vamb.taxvamb_encode.make_dataloader_concat_hloss": (vt.make_dataloader_concat_hloss, False),
            "vamb.taxvamb_encode.make_dataloader_labels_hloss": (vt.make_dataloader_labels_hloss, False),
            "vamb.taxvamb_encode.make_dataloader_semisupervised_hloss": (vt.make_dataloader_semisupervised_hloss, False),
            "vamb.taxvamb_encode.VAEVAEHLoss": (vt.VAEVAEHLoss.__init__, True), "VAEVAEHLoss.trainmodel": (vt.VAEVAEHLoss.trainmodel, True),
            "VAEJoint.encode": (vt.VAEConcatHLoss.encode, True), "vamb.cluster.ClusterGenerator": (vc.ClusterGenerator.__init__, True)}


def test_every_recorded_taxvamb_cli_call_binds_to_the_products_signatures():
    """Runs anywhere.  Every call the reference's run_vaevae makes on the names install(semisupervised=True) rebinds -- and the calls
    the reference's own loaders make on vamb.encode.make_dataloader from inside -- is a valid call of the product's function."""
    targets = _targets_taxvamb()
    seen = []
    for call in TRACE_TAX["trace"]:
        fn, is_method = targets[call["name"]]
        args = ([object()] if is_method else []) + list(call["args"])
        bound = inspect.signature(fn).bind(*args, **call["kwargs"])
        assert not any(k.startswith("_") for k in bound.arguments), "the CLI never passes a private parameter"
        seen.append(call["name"])
    assert set(seen) == set(targets), "the recorded run exercised every rebound name"
    # the CLI's own call order (vamb/__main__.py:1986, 1999, 2006, 2017, 2029, 2043, 2051, 2058); the two bare make_dataloader calls
    # come from inside the reference's concat / labels loaders
    outer = [c["name"] for c in TRACE_TAX["trace"] if not (c["name"] == "vamb.encode.make_dataloader" and not c["kwargs"])]
    assert outer == ["vamb.taxvamb_encode.VAEVAEHLoss", "vamb.encode.make_dataloader", "vamb.taxvamb_encode.make_dataloader_concat_hloss",
                     "vamb.taxvamb_encode.make_dataloader_labels_hloss", "vamb.taxvamb_encode.make_dataloader_semisupervised_hloss",
                     "VAEVAEHLoss.trainmodel", "VAEJoint.encode", "vamb.cluster.ClusterGenerator"]


@needs_reference
def test_real_taxvamb_cli_with_the_drop_in_bound_writes_the_reference_files(tmp_path):
    """``vamb bin taxvamb --no_predictor`` -- the reference's main(), option classes, run_vaevae -- with
    dropin.install(semisupervised=True) active.  No GPU here: the three networks stay the reference's (same seed, same threads: the
    golden run's latent, checked), the product's ClusterGenerator runs on the oracle backend inside the product's writer."""
    import cli_reference as cr
    import make_cli_golden as mk
    from oracle_backend import OracleScanBackend
    from vamb_amd import synth

    g = np.load(os.path.join(HERE, "golden", "cli_bin_taxvamb.npz"))
    c = TRACE_TAX["case"]
    comp, abundance, names, _ = cr.write_inputs(str(tmp_path), c["n"], c["nsamples"], c["data_seed"])
    assert list(names) == list(g["names"])
    _, _, _, labels = synth.features(c["n"], c["nsamples"], seed=c["data_seed"])
    assert cr.taxonomy_lines(names, labels) == [str(x) for x in g["taxonomy_lines"]]
    taxonomy = cr.write_taxonomy(str(tmp_path), names, labels)
    out = tmp_path / "out"
    r = cr.run_cli(mk.argv_for_taxvamb(c, out, comp, abundance, taxonomy, c["seed"]), binding="dropin", vae="reference",
                   backend_factory=OracleScanBackend, family="taxvamb")
    files = cr.read_outputs(out, prefix="vaevae", latent_name="vaevae_latent.npz")
    assert files["files"] == TRACE_TAX["files"]
    if not np.array_equal(files["latent"], g["latent"]):
        pytest.skip("the reference's CPU training did not reproduce the golden latent bit for bit on this host")
    assert files["vaevae_clusters_metadata.tsv"] == str(g["metadata_tsv"])
    assert files["vaevae_clusters_unsplit.tsv"] == str(g["unsplit_tsv"])
    a, b = files["vaevae_clusters_split.tsv"].splitlines(), str(g["split_tsv"]).splitlines()
    assert a[0] == b[0] and sorted(a) == sorted(b)

    def visible(lines):
        return [m for m in lines if "seconds" not in m and "Invoked with" not in m and ".npz" not in m and ".tsv" not in m
                and "Epoch:" not in m]

    assert visible(m for _, m in r["log"]) == visible(TRACE_TAX["log"])
