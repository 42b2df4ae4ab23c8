"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares (vambhip.h: the drop-in boundary;
vambhip_debug.h: test hooks and kernel diagnostics, kept out of it); the ctypes table in vamb_amd/_lib.py lists exactly the
same names.  No compute calls here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    so = os.path.join(ROOT, "vamb_amd", "libvambhip.so")
    if not os.path.exists(so):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "vamb_amd", "csrc", "build.py")])
    from vamb_amd import _lib

    return _lib.load()


HEADERS = ("vambhip.h", "vambhip_debug.h")


def header_symbols(names=HEADERS):
    found = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        found |= set(re.findall(r"\b(vh_[a-z_0-9]+)\s*\(", text))
    return sorted(found)


def test_debug_hooks_are_not_in_the_public_header():
    assert not [n for n in header_symbols(("vambhip.h",)) if n.startswith("vh_debug")]
    assert all(n.startswith("vh_debug") for n in header_symbols(("vambhip_debug.h",)))


def test_header_matches_ctypes_table(built_lib):
    from vamb_amd import _lib

    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_every_symbol_is_exported(built_lib):
    for name in header_symbols():
        assert hasattr(built_lib, name), name
    assert built_lib.vh_version().startswith(b"vambhip")


def test_invalid_arguments_become_valueerror(built_lib):
    import ctypes

    from vamb_amd import _lib

    # argument validation happens before any device work, so this runs without a GPU
    with pytest.raises(ValueError):
        _lib.check(built_lib.vh_clu_create(None, None, 0, 0, 0, None, ctypes.byref(ctypes.c_void_p())))
    assert b"NULL" in built_lib.vh_last_error() or b"observation" in built_lib.vh_last_error()


def test_edge_table_in_library_source_is_torch_linspace():
    import torch

    src = open(os.path.join(ROOT, "vamb_amd", "csrc", "cluster.hip")).read()
    block = src[src.index("c_edge_bits[VH_NBINS + 1] = {"):]
    block = block[: block.index("};")]
    bits = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", block)], dtype=np.uint32)
    assert len(bits) == 61
    assert np.array_equal(bits.view(np.float32), torch.linspace(0.0, 0.3, 61).numpy())


def test_no_gpu_fails_loudly(built_lib):
    from vamb_amd import _lib

    try:
        n = _lib.device_count()
    except _lib.VambHipError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is visible")
    from vamb_amd import cluster as vc

    with pytest.raises(_lib.VambHipError):
        vc.ClusterGenerator(np.ones((4, 3), np.float32), np.ones(4))


def test_header_is_plain_c(tmp_path):
    """include/vambhip.h is the C-ABI contract: it must compile as C99 (no C++ types, no torch types) and a C
    program that references every declared entry point must link against the library."""
    import os
    import re
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = "".join(open(os.path.join(root, "include", h)).read() for h in HEADERS)
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(vh_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 40
    src = tmp_path / "use_all.c"
    body = "\n".join(f"    p[{i}] = (fn)&{n};" for i, n in enumerate(names))
    src.write_text('#include "vambhip.h"\n#include "vambhip_debug.h"\ntypedef void (*fn)(void);\nint main(void) {\n    fn p[%d];\n%s\n'
                   '    return p[0] == p[1];\n}\n' % (len(names), body))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(root, "include"), str(src)])
    lib = os.path.join(root, "vamb_amd", "libvambhip.so")
    if os.path.exists(lib):
        out = tmp_path / "use_all"
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(out), lib,
                               "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined"])


def test_options_replace_environment_variables(monkeypatch):
    """The library reads no environment variable: options are set through vh_set_option, and the Python layer forwards the
    VAMBHIP_* variables when a handle is created (_lib.sync_env_options)."""
    from vamb_amd import _lib

    lib = _lib.load()
    assert _lib.get_option("scan.column_loop", 1) == 1          # unset: the caller's default comes back
    _lib.check(lib.vh_set_option(b"scan.column_loop", 0))
    assert _lib.get_option("scan.column_loop", 1) == 0
    _lib.check(lib.vh_unset_option(b"scan.column_loop"))
    assert _lib.get_option("scan.column_loop", 7) == 7
    monkeypatch.setenv("VAMBHIP_SPEC_WINDOW", "12")
    monkeypatch.setenv("VAMBHIP_NO_SPECULATION", "1")
    monkeypatch.delenv("VAMBHIP_SCAN_MFMA", raising=False)
    _lib.sync_env_options()
    assert _lib.get_option("gen.spec_window", -1) == 12 and _lib.get_option("gen.speculate", 1) == 0
    assert _lib.get_option("scan.mfma", 1) == 1
    monkeypatch.delenv("VAMBHIP_SPEC_WINDOW")
    monkeypatch.delenv("VAMBHIP_NO_SPECULATION")
    _lib.sync_env_options()
    assert _lib.get_option("gen.spec_window", -1) == -1 and _lib.get_option("gen.speculate", 1) == 1
    # no getenv left in the native sources
    import glob, os

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vamb_amd", "csrc")
    for path in glob.glob(os.path.join(root, "*.h*")):
        assert "getenv" not in open(path).read(), path
