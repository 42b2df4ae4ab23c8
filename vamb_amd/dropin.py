"""Bind the MI355X implementations onto an already-imported ``vamb`` package.

``vamb.__main__.trainvae`` and ``cluster_and_write_files`` look up ``vamb.encode.VAE`` /
``vamb.encode.make_dataloader`` / ``vamb.cluster.ClusterGenerator`` by attribute at call time
(``vamb/__main__.py:1075,1277,1458``), so replacing the attributes is enough for ``vamb bin default`` to
run on the GPU path unchanged.  The reference's own classes stay importable under ``*_reference`` names;
``vamb.semisupervised_encode`` captured the original ``VAE`` base class at import time and is unaffected.

    import vamb, vamb_amd.dropin
    vamb_amd.dropin.install()          # or: install(vamb)
    vamb.__main__.main()
"""
from __future__ import annotations

import sys


def install(vamb_module=None):
    """Replace the hot-path entry points of ``vamb`` by the ``vamb_amd`` ones.  Returns the dict of
    original objects (pass it to ``uninstall``)."""
    from . import cluster as _cluster
    from . import encode as _encode

    if vamb_module is None:
        vamb_module = sys.modules.get("vamb")
        if vamb_module is None:
            import vamb as vamb_module  # noqa: F811
    enc, clu = vamb_module.encode, vamb_module.cluster
    original = dict(VAE=enc.VAE, make_dataloader=enc.make_dataloader, set_batchsize=enc.set_batchsize,
                    ClusterGenerator=clu.ClusterGenerator, Cluster=clu.Cluster)
    enc.VAE_reference, clu.ClusterGenerator_reference = enc.VAE, clu.ClusterGenerator
    enc.VAE = _encode.VAE
    enc.make_dataloader = _encode.make_dataloader
    enc.set_batchsize = _encode.set_batchsize
    clu.ClusterGenerator = _cluster.ClusterGenerator
    clu.Cluster = _cluster.Cluster
    return original


def uninstall(original, vamb_module=None):
    if vamb_module is None:
        vamb_module = sys.modules["vamb"]
    enc, clu = vamb_module.encode, vamb_module.cluster
    enc.VAE, enc.make_dataloader, enc.set_batchsize = (original["VAE"], original["make_dataloader"],
                                                       original["set_batchsize"])
    clu.ClusterGenerator, clu.Cluster = original["ClusterGenerator"], original["Cluster"]
