#!/usr/bin/env python
"""Instruction statistics of one kernel in a hipcc -S dump (developer tool, not product code).

    python tools/isa_stats.py vae.s 'gemm_bf16_kernelILi128ELi128ELi2ELi4ELi3ELi0'

Prints register / LDS usage from the kernel descriptor comments and an instruction histogram per region
(regions are split at s_barrier instructions so prologue / K loop / epilogue can be told apart)."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and pat in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while not lines[end].startswith("\t.section") and "s_endpgm" not in lines[end]:
        end += 1
    # keep going to the .end_amdhsa_kernel for metadata
    meta_end = end
    while meta_end < len(lines) and ".end_amdhsa_kernel" not in lines[meta_end]:
        meta_end += 1
    body = lines[start:end + 1]
    print(lines[start])
    for l in lines[end:meta_end + 40]:
        if re.search(r"; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|codeLenInByte)", l):
            print("  ", l.strip())
    regions = [collections.Counter()]
    for l in body:
        l = l.strip()
        if not l or l.startswith((";", ".", "_Z")) or l.endswith(":"):
            continue
        op = l.split()[0]
        regions[-1][op] += 1
        if op == "s_barrier":
            regions.append(collections.Counter())

    def cls(op):
        if op.startswith("v_mfma"):
            return "mfma"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        if op.startswith("v_"):
            return "valu"
        if op.startswith("s_waitcnt"):
            return "wait"
        if op.startswith("s_"):
            return "salu"
        return "other"

    for ri, r in enumerate(regions):
        c = collections.Counter()
        for op, n in r.items():
            c[cls(op)] += n
        print(f"region {ri}: total {sum(r.values())}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if "-v" in sys.argv:
            for op, n in r.most_common(25):
                print(f"      {n:6d} {op}")


if __name__ == "__main__":
    main()
