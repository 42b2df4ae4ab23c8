"""Bind the MI355X implementations onto an already-imported ``vamb`` package.

``vamb.__main__.trainvae`` and ``cluster_and_write_files`` look up ``vamb.encode.VAE`` /
``vamb.encode.make_dataloader`` / ``vamb.cluster.ClusterGenerator`` by attribute at call time
(``vamb/__main__.py:1075,1277,1458``), so replacing the attributes is enough for ``vamb bin default`` to
run on the GPU path unchanged.  The reference's own classes stay importable under ``*_reference`` names;
``vamb.semisupervised_encode`` captured the original ``VAE`` base class at import time and is unaffected.
``Composition._project`` (the TNF projection, row N2) and ``cluster_and_write_files`` (row N3) are rebound too; the
submodules are imported on demand and a hook that cannot be installed is reported with a ``RuntimeWarning``.

    import vamb, vamb_amd.dropin
    vamb_amd.dropin.install()          # or: install(vamb)
    vamb.__main__.main()
"""
from __future__ import annotations

import sys


def install(vamb_module=None, semisupervised: bool = False, strict: bool = False):
    """Replace the hot-path entry points of ``vamb`` by the ``vamb_amd`` ones.  Returns the dict of
    original objects (pass it to ``uninstall``).

    ``semisupervised=True`` also rebinds the TaxVamb training side (row N4): ``vamb.semisupervised_encode.{VAELabels, VAEConcat,
    VAEVAE, make_dataloader_labels, make_dataloader_concat}`` and ``vamb.taxvamb_encode.{VAELabelsHLoss, VAEConcatHLoss,
    VAEVAEHLoss, make_dataloader_labels_hloss, make_dataloader_concat_hloss, make_dataloader_semisupervised_hloss}`` -- the names
    ``vamb bin taxvamb`` looks up at call time (``vamb/__main__.py:1988-2047``).  ``vamb.taxvamb_encode`` is imported BEFORE the
    rebinding, while ``vamb.semisupervised_encode``'s classes are still the torch ones: its remaining classes (Taxometer's
    ``VAMB2Label``) keep their torch bases.  TaxVamb's clustering is on the GPU either way (``cluster_and_write_files`` below).

    ``strict=True``: a submodule of ``vamb`` that cannot be imported (``vamb.parsecontigs`` needs ``vambcore``, say) raises
    instead of leaving that hook on the reference's path with a ``RuntimeWarning`` naming the cause."""
    global _STRICT
    _STRICT = bool(strict)
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
    # (parsecontigs.py:155): every 1000 contigs' worth of raw 4-mer counts is projected on the GPU.  The submodule is
    # imported here if the caller has not done so yet (a hook that is silently skipped leaves the slow path running).
    pc = _submodule(vamb_module, "parsecontigs")
    if pc is not None and hasattr(pc, "Composition"):
        original["_project"] = pc.Composition.__dict__["_project"]
        pc.Composition._project = staticmethod(_make_project(pc._KERNEL))
    else:
        _warn("vamb.parsecontigs.Composition not found: the TNF projection stays on the reference's numpy path")
    # row N3: the output loop of `vamb bin default` (vamb/__main__.py:1254-1404); its callers look the name up in the
    # module.  Under `python -m vamb` that module is registered as `__main__`, under an entry-point script as
    # `vamb.__main__`: every module object that is vamb's __main__ is patched.
    from . import output as _output

    mains = _main_modules(vamb_module)
    original["cluster_and_write_files"] = [(m, m.cluster_and_write_files) for m in mains]
    for m in mains:
        m.cluster_and_write_files = _output.cluster_and_write_files
    if not mains:
        _warn("vamb.__main__ is not importable: cluster_and_write_files stays the reference's")
    if semisupervised:
        ss = _submodule(vamb_module, "semisupervised_encode")
        if ss is None:
            _warn("vamb.semisupervised_encode is not importable: VAELabels / VAEConcat / VAEVAE stay the reference's")
        else:
            from . import semisupervised_encode as _ss

            # (classes of vamb.taxvamb_encode subclass ss.VAELabels / ss.VAEConcat: import it while those are the torch classes)
            tx = _submodule(vamb_module, "taxvamb_encode")
            names = ("VAELabels", "VAEConcat", "VAEVAE", "make_dataloader_labels", "make_dataloader_concat")
            original["semisupervised"] = {n: getattr(ss, n) for n in names}
            for n in names:
                setattr(ss, n, getattr(_ss, n))
            if tx is None:
                _warn("vamb.taxvamb_encode is not importable: nothing to rebind for TaxVamb's trainer")
            else:
                from . import taxvamb_encode as _tx

                tnames = ("VAELabelsHLoss", "VAEConcatHLoss", "VAEVAEHLoss", "make_dataloader_labels_hloss",
                          "make_dataloader_concat_hloss", "make_dataloader_semisupervised_hloss")
                original["taxvamb"] = {n: getattr(tx, n) for n in tnames}
                for n in tnames:
                    setattr(tx, n, getattr(_tx, n))
    return original


def _warn(msg: str) -> None:
    import warnings

    warnings.warn("vamb_amd.dropin: " + msg, RuntimeWarning, stacklevel=3)


_STRICT = False


def _submodule(vamb_module, name: str):
    """``vamb.<name>``, imported on demand.  Only a failed IMPORT (a missing dependency such as vambcore, a stub package in the
    tests) is tolerated, and its cause is reported; any other exception raised while the module executes is a bug and
    propagates.  ``install(strict=True)`` re-raises the ImportError as well."""
    mod = getattr(vamb_module, name, None)
    if mod is None:
        import importlib

        try:
            mod = importlib.import_module(vamb_module.__name__ + "." + name)
        except ImportError as exc:
            if _STRICT:
                raise
            _warn(f"{vamb_module.__name__}.{name} could not be imported ({exc!r})")
            return None
    return mod


def _main_modules(vamb_module):
    """Every module object that is vamb's ``__main__``: ``vamb.__main__`` (imported on demand) and, under
    ``python -m vamb``, the running ``__main__`` itself (importing ``vamb.__main__`` then would run the CLI twice)."""
    import os

    pkg_dirs = {os.path.abspath(str(p)) for p in getattr(vamb_module, "__path__", [])}

    def is_vambs_main(m) -> bool:
        f = getattr(m, "__file__", None)
        return bool(f) and os.path.basename(f).startswith("__main__") and os.path.dirname(os.path.abspath(f)) in pkg_dirs

    found = []
    running = sys.modules.get("__main__")
    if running is not None and is_vambs_main(running):
        found.append(running)
    named = sys.modules.get(vamb_module.__name__ + ".__main__")
    if named is None and not found:
        named = _submodule(vamb_module, "__main__")
    if named is not None and named not in found:
        found.append(named)
    return [m for m in found if hasattr(m, "cluster_and_write_files")]


_MAX_PROJECTORS = 4


def _make_project(default_kernel):
    """Composition._project with the reference's signature ``(fourmers, kernel=_KERNEL)``.

    One device projector per kernel CONTENT (shape + bytes): ``id(kernel)`` is reused by CPython once an array is
    collected and says nothing about an in-place edit.  The cache is bounded; an evicted projector releases its stream
    and device buffers."""
    import collections

    import numpy as _np

    from . import composition as _composition

    projectors = collections.OrderedDict()

    def _project(fourmers, kernel=default_kernel):
        k = _np.ascontiguousarray(kernel, dtype=_np.float32)
        key = (k.shape, k.tobytes())   # 105 KB: negligible next to the [n x 256] counts of a call
        proj = projectors.get(key)
        if proj is None:
            proj = _composition.TnfProjector(k)
            projectors[key] = proj
            while len(projectors) > _MAX_PROJECTORS:
                _, old = projectors.popitem(last=False)
                old.close()
        else:
            projectors.move_to_end(key)
        return proj.project(fourmers)

    _project._projectors = projectors   # introspection (tests)
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
    for m, fn in original.get("cluster_and_write_files", []):
        m.cluster_and_write_files = fn
    for n, obj in original.get("semisupervised", {}).items():
        setattr(vamb_module.semisupervised_encode, n, obj)
    for n, obj in original.get("taxvamb", {}).items():
        setattr(vamb_module.taxvamb_encode, n, obj)
