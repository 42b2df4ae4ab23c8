"""Bind the MI355X implementations onto an already-imported ``vamb`` package.

``vamb.__main__.trainvae`` and ``cluster_and_write_files`` look up ``vamb.encode.VAE`` /
``vamb.encode.make_dataloader`` / ``vamb.cluster.ClusterGenerator`` by attribute at call time
(``vamb/__main__.py:1075,1277,1458``), so replacing the attributes is enough for ``vamb bin default`` to
run on the GPU path unchanged.  The reference's own classes stay importable under ``*_reference`` names;
``vamb.semisupervised_encode`` captured the original ``VAE`` base class at import time and is unaffected.
When ``vamb.parsecontigs`` is imported, ``Composition._project`` (the TNF projection, row N2) is rebound too.

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
    # row N2: Composition._project (vamb/parsecontigs.py:140-150) is looked up on the class by _convert at call time
    # (parsecontigs.py:155): every 1000 contigs' worth of raw 4-mer counts is projected on the GPU
    pc = getattr(vamb_module, "parsecontigs", None)
    if pc is not None and hasattr(pc, "Composition"):
        original["_project"] = pc.Composition.__dict__["_project"]
        pc.Composition._project = staticmethod(_make_project(pc._KERNEL))
    # row N3: the output loop of `vamb bin default` (vamb/__main__.py:1254-1404); its callers look the name up in the module
    main = sys.modules.get(vamb_module.__name__ + ".__main__")
    if main is not None and hasattr(main, "cluster_and_write_files"):
        from . import output as _output

        original["cluster_and_write_files"] = main.cluster_and_write_files
        main.cluster_and_write_files = _output.cluster_and_write_files
    return original


def _make_project(default_kernel):
    """Composition._project with the reference's signature ``(fourmers, kernel=_KERNEL)``; one device projector per kernel."""
    from . import composition as _composition

    projectors = {}

    def _project(fourmers, kernel=default_kernel):
        key = id(kernel)
        if key not in projectors:
            projectors[key] = _composition.TnfProjector(kernel)
        return projectors[key].project(fourmers)

    return _project


def uninstall(original, vamb_module=None):
    if vamb_module is None:
        vamb_module = sys.modules["vamb"]
    enc, clu = vamb_module.encode, vamb_module.cluster
    enc.VAE, enc.make_dataloader, enc.set_batchsize = (original["VAE"], original["make_dataloader"],
                                                       original["set_batchsize"])
    clu.ClusterGenerator, clu.Cluster = original["ClusterGenerator"], original["Cluster"]
    if "_project" in original:
        vamb_module.parsecontigs.Composition._project = original["_project"]
    if "cluster_and_write_files" in original:
        sys.modules[vamb_module.__name__ + ".__main__"].cluster_and_write_files = original["cluster_and_write_files"]
