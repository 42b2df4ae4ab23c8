"""vamb_amd.dropin binds our classes onto the reference package (build container only: needs
/root/reference; skipped on the GPU box)."""
import pytest

import ref_harness


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")
def test_install_and_uninstall():
    import sys

    ref_harness.load_reference()
    vamb = sys.modules["vamb"]
    from vamb_amd import cluster as vc, dropin, encode as ve

    ref_vae, ref_cg = vamb.encode.VAE, vamb.cluster.ClusterGenerator
    saved = dropin.install(vamb)
    try:
        assert vamb.encode.VAE is ve.VAE and vamb.encode.make_dataloader is ve.make_dataloader
        assert vamb.cluster.ClusterGenerator is vc.ClusterGenerator and vamb.cluster.Cluster is vc.Cluster
        assert vamb.encode.VAE_reference is ref_vae and vamb.cluster.ClusterGenerator_reference is ref_cg
        # same call signatures as the reference (names and defaults)
        import inspect

        for ours, theirs in ((ve.VAE.__init__, ref_vae.__init__), (ve.VAE.trainmodel, ref_vae.trainmodel),
                             (ve.VAE.encode, ref_vae.encode), (ve.make_dataloader, saved["make_dataloader"]),
                             (ve.set_batchsize, saved["set_batchsize"])):
            po = [(p.name, p.default) for p in inspect.signature(ours).parameters.values() if not p.name.startswith("_")]
            pt = [(p.name, p.default) for p in inspect.signature(theirs).parameters.values()]
            assert po == pt, (ours, po, pt)
        po = [(p.name, p.default) for p in inspect.signature(vc.ClusterGenerator.__init__).parameters.values()
              if not p.name.startswith("_")]
        pt = [(p.name, p.default) for p in inspect.signature(ref_cg.__init__).parameters.values()]
        assert po == pt
    finally:
        dropin.uninstall(saved, vamb)
    assert vamb.encode.VAE is ref_vae and vamb.cluster.ClusterGenerator is ref_cg


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")
def test_semisupervised_classes_mirror_the_reference():
    import torch
    """Row N4: VAELabels / VAEConcat and their loaders keep the reference's signatures (names and defaults;
    semisupervised_encode.py:111-175, 189-436, 438-698), write state_dicts of the reference's names and shapes, and
    dropin.install(semisupervised=True) binds them."""
    import inspect
    import sys
    import types

    ss = ref_harness.load_reference_module("semisupervised_encode")
    vamb = sys.modules["vamb"]
    from vamb_amd import dropin, semisupervised_encode as vs

    def sig(f):
        return [(p.name, p.default) for p in inspect.signature(f).parameters.values() if not p.name.startswith("_")]

    for cls in ("VAELabels", "VAEConcat"):
        for meth in ("__init__", "forward", "calc_loss", "trainepoch", "trainmodel", "encode"):
            assert sig(getattr(getattr(vs, cls), meth)) == sig(getattr(getattr(ss, cls), meth)), (cls, meth)
    for fn in ("make_dataloader_labels", "make_dataloader_concat", "collate_fn_labels", "collate_fn_concat"):
        assert sig(getattr(vs, fn)) == sig(getattr(ss, fn)), fn
    # state_dict names / shapes
    ref = ss.VAEConcat(6, 130, nhiddens=[48, 40], nlatent=8).state_dict()
    me = types.SimpleNamespace(nsamples=6, _native_nsamples=6, nlabels=130, _NL=130, nhiddens=[48, 40], nlatent=8)
    me._row_width = lambda: vs.VAEConcat._row_width(me)
    assert vs.VAEConcat._state_names(me) == list(ref.keys())
    for k in ref:
        assert tuple(vs.VAEConcat._shape_of(me, k)) == tuple(ref[k].shape), k
    ref = ss.VAELabels(150, nhiddens=[64, 32], nlatent=6).state_dict()
    me = types.SimpleNamespace(nsamples=46, nlabels=150, _NL=150, nhiddens=[64, 32], nlatent=6)
    me._row_width = lambda: vs.VAELabels._row_width(me)
    for k in ref:
        assert tuple(vs.VAELabels._shape_of(me, k)) == tuple(ref[k].shape), k
    # the joint trainer and the HLoss classes (taxvamb_encode.py:277-743): same constructor / method signatures
    tx = ref_harness.load_reference_module("taxvamb_encode")
    from vamb_amd import taxvamb_encode as vt

    for cls, meths in (("VAELabelsHLoss", ("__init__", "calc_loss", "trainmodel")), ("VAEConcatHLoss", ("__init__", "calc_loss")),
                       ("VAEVAEHLoss", ("__init__", "calc_loss_joint", "trainepoch", "trainmodel", "save", "load"))):
        for meth in meths:
            assert sig(getattr(getattr(vt, cls), meth)) == sig(getattr(getattr(tx, cls), meth)), (cls, meth)
    for fn in ("make_dataloader_labels_hloss", "make_dataloader_concat_hloss", "make_dataloader_semisupervised_hloss",
               "permute_indices", "make_graph", "kld_gauss", "collate_fn_labels_hloss", "collate_fn_concat_hloss",
               "collate_fn_semisupervised_hloss"):
        assert sig(getattr(vt, fn)) == sig(getattr(tx, fn)), fn
    assert sig(vs.VAEVAE.__init__) == sig(ss.VAEVAE.__init__) and sig(vs.VAEVAE.trainmodel) == sig(ss.VAEVAE.trainmodel)
    assert vs.VAEVAE_METRICS[0] == "loss_vamb" and len(vs.VAEVAE_METRICS) == 17
    theirs, their_joint = ss.VAELabels, tx.VAEVAEHLoss
    saved = dropin.install(vamb, semisupervised=True)
    try:
        assert ss.VAELabels is vs.VAELabels and ss.VAEConcat is vs.VAEConcat and ss.VAEVAE is vs.VAEVAE
        assert ss.make_dataloader_concat is vs.make_dataloader_concat
        # what `vamb bin taxvamb` looks up at call time (__main__.py:1988-2047) is the GPU trainer and its loaders
        assert tx.VAEVAEHLoss is vt.VAEVAEHLoss and tx.VAELabelsHLoss is vt.VAELabelsHLoss
        assert tx.make_dataloader_semisupervised_hloss is vt.make_dataloader_semisupervised_hloss
        assert issubclass(tx.VAMB2Label, torch.nn.Module)   # Taxometer's classifier is untouched
    finally:
        dropin.uninstall(saved, vamb)
    assert ss.VAELabels is theirs and tx.VAEVAEHLoss is their_joint


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")
def test_semisupervised_loaders_yield_the_reference_batches():
    """make_dataloader_concat / make_dataloader_labels (semisupervised_encode.py:111-175): same dataset tensors, same one-hot
    batches from the collate functions, same batch size / drop_last as the reference's loaders on the same inputs."""
    import numpy as np
    import torch

    ss = ref_harness.load_reference_module("semisupervised_encode")
    from vamb_amd import encode as ve, semisupervised_encode as vs, synth

    ab, tnf, lens, genome = synth.features(300, 5, seed=11, k=7)
    labels = np.array([f"g{int(i)}" for i in genome])
    ve.set_prep_mode("host")
    try:
        ours = vs.make_dataloader_concat(ab.copy(), tnf.copy(), lens, labels, batchsize=64)
    finally:
        ve.set_prep_mode("auto")
    theirs = ss.make_dataloader_concat(ab.copy(), tnf.copy(), lens, labels, batchsize=64)
    assert ours.batch_size == theirs.batch_size and ours.drop_last == theirs.drop_last
    for a, b in zip(ours.dataset.tensors, theirs.dataset.tensors):
        assert torch.equal(a, b)
    rows = [ours.dataset[i] for i in range(10)]
    got, want = ours.collate_fn(rows), theirs.collate_fn([theirs.dataset[i] for i in range(10)])
    assert len(got) == len(want) == 5
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and torch.equal(a, b)
    assert got[4].shape == (10, 105)
    ours_l = vs.make_dataloader_labels(ab, tnf, lens, labels, batchsize=64)
    theirs_l = ss.make_dataloader_labels(ab.copy(), tnf.copy(), lens, labels, batchsize=64)
    assert torch.equal(ours_l.dataset.tensors[0], theirs_l.dataset.tensors[0])
    a = ours_l.collate_fn([ours_l.dataset[i] for i in range(7)])
    b = theirs_l.collate_fn([theirs_l.dataset[i] for i in range(7)])
    assert len(a) == len(b) == 1 and torch.equal(a[0], b[0])


@pytest.mark.reference
def test_state_dict_spec_matches_reference():
    """The names, order and shapes our VAE.state_dict()/save() writes (VAE._state_names / _shape_of) are exactly those
    of the reference module's state_dict (encode.py:226-249, 486-502), for several architectures -- so a model.pt
    written here passes the reference's strict load_state_dict and vice versa.  (The end-to-end check -- the
    reference's VAE.load reading a file written on the GPU -- is oracle/check_model_pt.py, profiles/r02_model_pt_crossload.json.)"""
    import types

    import ref_harness
    from vamb_amd import encode as ve

    if not ref_harness.reference_available():
        pytest.skip("needs /root/reference")
    _, _, ref_encode = ref_harness.load_reference()
    for nsamples, nhiddens, nlatent in [(6, [48, 40], 8), (1, [24, 24], 4), (7, [32, 24, 16], 6), (50, [512, 512], 32)]:
        ref = ref_encode.VAE(nsamples, nhiddens=list(nhiddens), nlatent=nlatent).state_dict()
        me = types.SimpleNamespace(nsamples=nsamples, nhiddens=list(nhiddens), nlatent=nlatent)
        me._row_width = lambda: ve.VAE._row_width(me)
        names = ve.VAE._state_names(me)
        assert names == list(ref.keys())
        for k in names:
            assert tuple(ve.VAE._shape_of(me, k)) == tuple(ref[k].shape), k


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")
def test_install_rebinds_the_output_writer_and_the_tnf_projection():
    """dropin.install also rebinds vamb.__main__.cluster_and_write_files (row N3: every binner of the reference -- default,
    taxvamb, avamb -- writes its clusters through that one function, vamb/__main__.py:1266,1523,2050) and
    Composition._project (row N2) when those modules are loaded.  vamb/__main__.py cannot be imported in this image, so a
    stand-in module object carries the attribute."""
    import importlib.util
    import os
    import sys
    import types

    ref_harness.load_reference()
    vamb = sys.modules["vamb"]
    from vamb_amd import dropin, output

    spec = importlib.util.spec_from_file_location("vamb.parsecontigs",
                                                  os.path.join(ref_harness.REFERENCE_ROOT, "vamb", "parsecontigs.py"))
    pc = importlib.util.module_from_spec(spec)
    sys.modules["vamb.parsecontigs"] = pc
    spec.loader.exec_module(pc)
    vamb.parsecontigs = pc
    main = types.ModuleType("vamb.__main__")
    main.cluster_and_write_files = object()
    sys.modules["vamb.__main__"] = main
    ref_project = pc.Composition.__dict__["_project"]
    saved = dropin.install(vamb)
    try:
        assert main.cluster_and_write_files is output.cluster_and_write_files
        assert pc.Composition.__dict__["_project"] is not ref_project
        import inspect

        ours = [(p.name, p.default) for p in inspect.signature(pc.Composition._project).parameters.values()]
        theirs = [(p.name, p.default is not inspect.Parameter.empty) for p in inspect.signature(ref_project.__func__).parameters.values()]
        assert [n for n, _ in ours] == [n for n, _ in theirs] == ["fourmers", "kernel"]
    finally:
        dropin.uninstall(saved, vamb)
        del sys.modules["vamb.__main__"]
    assert pc.Composition.__dict__["_project"] is ref_project
