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
            po = [(p.name, p.default) for p in inspect.signature(ours).parameters.values()]
            pt = [(p.name, p.default) for p in inspect.signature(theirs).parameters.values()]
            assert po == pt, (ours, po, pt)
        po = [(p.name, p.default) for p in inspect.signature(vc.ClusterGenerator.__init__).parameters.values()
              if not p.name.startswith("_")]
        pt = [(p.name, p.default) for p in inspect.signature(ref_cg.__init__).parameters.values()]
        assert po == pt
    finally:
        dropin.uninstall(saved, vamb)
    assert vamb.encode.VAE is ref_vae and vamb.cluster.ClusterGenerator is ref_cg
