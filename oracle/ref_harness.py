"""TEST INFRASTRUCTURE ONLY -- loads the *real* reference hot-path modules as an oracle.

This file is part of ``oracle/``: only ``tests/``, ``tests/golden/make_golden.py``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product
(``vamb_amd/``) never does.

It imports ``/root/reference/vamb/{vambtools,cluster,encode}.py`` *by path* under a synthetic
``vamb`` package (the packaged ``vamb/__init__.py:15-31`` cannot import here: it pulls in
pycoverm/pyhmmer/pyrodigal and ``importlib.metadata.version("vamb")``).  Three third-party modules
that are absent from this image are replaced by stubs:

* ``loguru.logger``            -> no-op object (logging only; ``encode.py:11,427``)
* ``vambcore.overwrite_matrix`` -> restated below from its contract, which the reference pins in
  ``test/test_vambtools.py:271-298`` (result == ``arr[mask]``, returns the kept-row count) and uses at
  ``vamb/vambtools.py:302,319``.  ``vambcore.kmercounts`` is off-path and raises.
* ``dadaptation.DAdaptAdam``    -> ``oracle/dadapt_restated.py`` (dadaptation==3.2, pinned in
  ``pyproject.toml:12``; PARITY UNPINNED -- the package source is not in this image).

``/root/reference`` exists only in the build container, never on the GPU box: callers must use
``reference_available()`` and skip otherwise.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("VAMB_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "vamb", "cluster.py"))


class _NullLogger:
    def __getattr__(self, name):
        def _noop(*args, **kwargs):
            return self

        return _noop

    def catch(self, *args, **kwargs):  # decorator form used in __main__, harmless here
        def deco(fn):
            return fn

        return deco


def _overwrite_matrix(arr: np.ndarray, mask: np.ndarray) -> int:
    """Restatement of vambcore.overwrite_matrix (Rust): order-preserving in-place row compaction.

    Contract (reference ``vamb/vambtools.py:291-321`` and ``test/test_vambtools.py:271-298``): after
    the call the first ``n`` rows of ``arr`` equal ``arr_before[mask]`` and ``n`` is returned.
    """
    mask = np.asarray(mask, dtype=bool)
    if len(mask) != len(arr):
        raise ValueError("Lengths of array and mask must match")
    kept = np.flatnonzero(mask)
    n = len(kept)
    if n:
        # fancy-index read materialises a copy first, so the overlapping write is safe
        arr[:n] = arr[kept]
    return n


def _kmercounts(*args, **kwargs):  # pragma: no cover - off the hot path
    raise NotImplementedError("vambcore.kmercounts is outside the hot path (SURVEY.md section 2, row 5)")


_cached = None


def load_reference():
    """Return ``(vambtools, cluster, encode)`` -- the reference's own modules, executed unmodified."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import dadapt_restated  # noqa: E402  (oracle/dadapt_restated.py)

    if "loguru" not in sys.modules:
        loguru = types.ModuleType("loguru")
        loguru.logger = _NullLogger()
        sys.modules["loguru"] = loguru
    if "vambcore" not in sys.modules:
        vambcore = types.ModuleType("vambcore")
        vambcore.overwrite_matrix = _overwrite_matrix
        vambcore.kmercounts = _kmercounts
        sys.modules["vambcore"] = vambcore
    if "dadaptation" not in sys.modules:
        dad = types.ModuleType("dadaptation")
        dad.DAdaptAdam = dadapt_restated.DAdaptAdam
        sys.modules["dadaptation"] = dad

    pkg_name = "vamb"
    if pkg_name in sys.modules and not getattr(sys.modules[pkg_name], "__oracle_stub__", False):
        raise RuntimeError("a real 'vamb' package is already imported; refusing to shadow it")
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "vamb")]
    pkg.__oracle_stub__ = True
    sys.modules[pkg_name] = pkg

    mods = []
    for name in ("vambtools", "cluster", "encode"):
        full = f"{pkg_name}.{name}"
        spec = importlib.util.spec_from_file_location(
            full, os.path.join(REFERENCE_ROOT, "vamb", f"{name}.py")
        )
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        mods.append(mod)
    _cached = tuple(mods)
    return _cached


def load_reference_module(name: str):
    """One more module of the reference package, executed unmodified (after load_reference()): e.g.
    ``semisupervised_encode`` (imports only vamb.encode, vamb.vambtools, torch and loguru)."""
    load_reference()
    full = f"vamb.{name}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, os.path.join(REFERENCE_ROOT, "vamb", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    setattr(sys.modules["vamb"], name, mod)
    return mod
