#!/usr/bin/env python
"""Build a VARIANT of libvambhip.so beside the product library, for same-box A/B runs (VAMBHIP_LIB_PATH=vamb_amd/libvambhip_<name>.so).

    python tools/build_variant.py <name> [--src cluster.hip[,vae.hip]] [-DFLAG ...]

The named sources are recompiled with the extra flags into vamb_amd/csrc/build/<name>/; every other object is the product build's.
`timing` is the conventional name of the build with -DVAMBHIP_TIMING_EXPERIMENTS (scan.debug switches, vh_debug_scan_timeline).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vamb_amd", "csrc"))
import build as pb  # noqa: E402


def main():
    name = sys.argv[1]
    srcs = ["cluster.hip"]
    flags = []
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--src":
            srcs = args.pop(0).split(",")
        else:
            flags.append(a)
    pb.build(verbose=False)
    out_dir = os.path.join(pb.OBJ_DIR, name)
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src, extra in pb.SOURCES.items():
        obj = os.path.join(pb.OBJ_DIR, src.replace(".hip", ".o"))
        if src in srcs:
            obj = os.path.join(out_dir, src.replace(".hip", ".o"))
            cmd = [pb.hipcc(), *pb.COMMON, *extra, *flags, "-c", os.path.join(pb.HERE, src), "-o", obj]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(pb.PKG, f"libvambhip_{name}.so")
    subprocess.check_call([pb.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    print(out)


if __name__ == "__main__":
    main()
