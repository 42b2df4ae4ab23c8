"""TEST INFRASTRUCTURE ONLY -- imports the reference's UNMODIFIED ``vamb/__main__.py`` (the CLI) in the build container.

``vamb/__main__.py`` pulls in ``pycoverm`` (BAM parsing), ``pyhmmer`` / ``pyrodigal`` (marker genes, through
``vamb.parsemarkers`` / ``vamb.reclustering``) and ``importlib.metadata.version("vamb")`` (through ``vamb/__init__.py``); none of
them is in this image and none is on the hot path.  They are replaced by *attribute-bearing stubs*: module objects whose names
resolve (the reference uses them in annotations evaluated at ``def`` time: ``pyhmmer.plan7.HMM``, ``pyrodigal.GeneFinder``,
``pyhmmer.easel.DigitalSequence``) and whose callables raise ``NotImplementedError`` when CALLED.  Everything the default binner
executes between ``main()`` and the result files -- argument parsing, option classes, ``load_composition_and_abundance`` from
``.npz`` inputs, ``run_bin_default``, ``trainvae``, ``cluster_and_write_files`` -- is the reference's own code, byte for byte.

Used by ``tests/golden/make_golden.py`` (records the CLI golden: call trace, result files) and by ``tests/test_cli_dropin.py``.
``/root/reference`` exists only in the build container: callers check ``ref_harness.reference_available()``.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import ref_harness

# every submodule ``vamb/__init__.py:15-26`` imports, in its order
_SUBMODULES = ("vambtools", "parsebam", "parsecontigs", "parsemarkers", "taxonomy", "cluster", "encode", "aamb_encode",
               "semisupervised_encode", "hloss_misc", "taxvamb_encode", "reclustering")


class _Refuses:
    """Stands in for a class / function of an absent third-party package: usable as an annotation, raises when called."""

    def __init__(self, qualname: str):
        self._qualname = qualname

    def __call__(self, *args, **kwargs):
        raise NotImplementedError(f"{self._qualname} is not available in this image (off the hot path; stubbed by oracle/ref_main.py)")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Refuses(f"{self._qualname}.{name}")

    def __repr__(self):
        return f"<stub {self._qualname}>"


def _stub_module(name: str, attrs=()) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__oracle_stub__ = True
    for a in attrs:
        setattr(mod, a, _Refuses(f"{name}.{a}"))

    def _module_getattr(attr, _n=name):   # PEP 562: any other public attribute resolves too (inspect & co. probe dunders)
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Refuses(f"{_n}.{attr}")

    mod.__getattr__ = _module_getattr
    return mod


def _install_third_party_stubs() -> None:
    if "pycoverm" not in sys.modules:
        sys.modules["pycoverm"] = _stub_module("pycoverm", ("is_bam_sorted", "get_coverages_from_bam"))
    if "pyrodigal" not in sys.modules:
        sys.modules["pyrodigal"] = _stub_module("pyrodigal", ("GeneFinder",))
    if "pyhmmer" not in sys.modules:
        ph = _stub_module("pyhmmer", ("hmmsearch",))
        for sub, names in (("plan7", ("HMM", "HMMFile")), ("easel", ("Alphabet", "DigitalSequence", "TextSequence")),
                           ("hmmer", ())):
            m = _stub_module(f"pyhmmer.{sub}", names)
            setattr(ph, sub, m)
            sys.modules[f"pyhmmer.{sub}"] = m
        sys.modules["pyhmmer"] = ph
    # vambcore.kmercounts: the Rust k-mer counter is off this path's .npz inputs; the oracle's restatement keeps FASTA input usable
    vc = sys.modules.get("vambcore")
    if vc is not None and getattr(vc, "kmercounts", None) is ref_harness._kmercounts:
        try:
            import kmer_oracle

            if hasattr(kmer_oracle, "kmercounts"):
                vc.kmercounts = kmer_oracle.kmercounts
        except ImportError:
            pass


class RecordingLogger(ref_harness._NullLogger):
    """loguru stand-in that keeps the messages (``messages``): the CLI's log is part of what a user sees."""

    def __init__(self):
        self.messages = []

    def __getattr__(self, name):
        if name in ("info", "warning", "error", "debug", "success"):
            def _log(msg, *a, **k):
                self.messages.append((name, str(msg)))
                return self

            return _log
        return super().__getattr__(name)


_main = None


def load_reference_main():
    """``(vamb, vamb.__main__)``: the stub package of ``ref_harness`` completed with every submodule of the real package, and the
    CLI module executed unmodified under the name ``vamb.__main__`` (so its ``if __name__ == "__main__"`` block does not run)."""
    global _main
    if _main is not None:
        return sys.modules["vamb"], _main
    ref_harness.load_reference()
    _install_third_party_stubs()
    pkg = sys.modules["vamb"]
    for name in _SUBMODULES:
        ref_harness.load_reference_module(name)
    pkg.__version_str__ = "0+reference.tree"   # vamb/__init__.py:31 asks importlib.metadata, which knows no installed 'vamb'
    pkg.__all__ = list(_SUBMODULES)
    full = "vamb.__main__"
    spec = importlib.util.spec_from_file_location(full, os.path.join(ref_harness.REFERENCE_ROOT, "vamb", "__main__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    env_before = {k: os.environ.get(k) for k in ("MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "OMP_NUM_THREADS")}
    path_before = list(sys.path)
    try:
        spec.loader.exec_module(mod)
    finally:
        # __main__.py:40-47 exports thread counts and appends the reference tree to sys.path at import time: undone, so that
        # importing the CLI does not change how the rest of the test session runs
        for k, v in env_before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        sys.path[:] = path_before
    setattr(pkg, "__main__", mod)
    _main = mod
    return pkg, mod


def unload_reference_main() -> None:
    global _main
    sys.modules.pop("vamb.__main__", None)
    pkg = sys.modules.get("vamb")
    if pkg is not None and hasattr(pkg, "__main__"):
        delattr(pkg, "__main__")
    _main = None
