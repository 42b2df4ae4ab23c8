#!/usr/bin/env python
"""Build libvambhip.so (gfx950 only) in-tree with hipcc.  No cmake, no torch extension machinery:
the product is a plain C-ABI shared library (include/vambhip.h).

    python vamb_amd/csrc/build.py [--force]

cluster.hip is compiled with -ffp-contract=off (its arithmetic is bit-exact against the oracle);
the VAE kernels use the default (fast) contraction.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libvambhip.so")
OBJ_DIR = os.path.join(HERE, "build")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Wall", "-Wno-unused-function"]
SOURCES = {
    # -amdgpu-atomic-optimizer-strategy=None: the scan kernel's LDS atomics sit on a rare, sparsely populated path;
    # the wave-aggregation loops the optimizer wraps around each of them cost more than the few atomics they save
    # -fno-slp-vectorize: the SLP vectoriser pairs the fmaf chains of two rows into v_pk_fma_f32, which gfx950 issues at
    # a quarter of the v_fma_f32 rate per lane-op (measured: 17 cycles per v_pk_fma_f32 in the scan kernel); plain
    # v_fmac_f32 with a scalar query operand runs the chains at the VALU's 2 cycles per wavefront instruction
    # -pragma-unroll-threshold: the evaluation loops of the scan kernels (one drain site per (medoid, row slot) pair, up to 32 per
    # kernel, each with the reference-order re-evaluation inlined) exceed the default budget of `#pragma unroll` (16 K
    # instructions); left rolled they index their accumulators dynamically and spill to scratch
    "cluster.hip": ["-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
                    "-mllvm", "-pragma-unroll-threshold=400000"],
    "vae.hip": [],
    # prep.hip reproduces numpy's float32 results bit for bit: no fused multiply-add
    "prep.hip": ["-ffp-contract=off"],
    "tnf.hip": ["-ffp-contract=off"],
    "comm.hip": [],
    "selftest.hip": [],   # host code on top of the C ABI (start-up self-test of the hand-scheduled kernels)
}


# Sources whose kernels wait for inline-asm register loads with hand-written s_waitcnt (the deep-prefetch / K-group fp32 GEMM tiles,
# the row-major scan): compiled with -save-temps so that the ISA of exactly the object that ships is at hand, and checked by
# isa_pending_loads.py -- no instruction may touch a register a load is still writing.  A hazard fails the build.
ISA_CHECKED = ("cluster.hip", "vae.hip")
ISA_SUFFIX = "-hip-amdgcn-amd-amdhsa-gfx950.s"


def isa_path(src: str) -> str:
    return os.path.join(OBJ_DIR, src.replace(".hip", "") + ISA_SUFFIX)


def check_isa(verbose: bool = True) -> None:
    import isa_pending_loads as ipl
    for src in ISA_CHECKED:
        path = isa_path(src)
        stamp = path + ".checked"
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: rebuild with `python vamb_amd/csrc/build.py --force`")
        if os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(path):
            continue
        report = []
        n_k, n_asm, hazards = ipl.check_file(path, out=report.append)
        if verbose:
            print(f"ISA check {os.path.basename(path)}: {n_k} kernels, {n_asm} with hand-waited register loads, {hazards} hazard(s)", flush=True)
        if hazards:
            raise RuntimeError("a register is touched while an inline-asm load into it is in flight:\n" + "\n".join(report))
        open(stamp, "w").write("ok\n")


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "vambhip.h"))
    headers.append(os.path.join(os.path.dirname(PKG), "include", "vambhip_debug.h"))
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        spath = os.path.join(HERE, src)
        if not os.path.exists(spath):
            continue
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        stale = force or newer(spath, obj) or any(newer(h, obj) for h in headers) or newer(__file__, obj)
        temps = ["-save-temps=obj"] if src in ISA_CHECKED else []
        if stale or (temps and not os.path.exists(isa_path(src))):
            jobs.append([hipcc(), *COMMON, *extra, *temps, "-c", spath, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for f in os.listdir(OBJ_DIR):   # the bulky by-products of -save-temps (preprocessed sources, bitcode); the device ISA stays
        if f.endswith((".hipi", ".bc", ".hipfb", ".out", ".resolution.txt", "-host-x86_64-unknown-linux-gnu.s", "-gfx950.o")):
            os.remove(os.path.join(OBJ_DIR, f))
    sys.path.insert(0, HERE)
    check_isa(verbose)
    if jobs or force or not os.path.exists(OUT):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
