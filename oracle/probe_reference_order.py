#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (oracle/; build container: needs torch CPU and /root/reference).

How the REFERENCE evaluates the two floating-point reductions of the cluster path on this container's torch build
(2.10 / oneMKL 2024.2 / AVX-512) -- `matrix.matmul(matrix[index])` (cluster.py:674) and `matrix.norm(dim=1)` (cluster.py:668):

1. probe: for an anchor column i holding 1.0 and two columns j, k holding 2^-24 (half an ulp of 1), the float32 result is
   1 + 2^-23 iff j and k are added to each other before either meets i.  All (i, j, k) reveal the summation tree.
2. confirm: the order read off the probe, restated in oracle/cluster_scan.c (vo_set_order(1)), reproduces torch's distances and
   the reference's `_normalize` bit for bit on random matrices of every latent width 1..257 and on the golden fixtures' inputs.

    python oracle/probe_reference_order.py > profiles/r03_reference_order_probe.txt
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cluster_oracle as co  # noqa: E402
import fixture_defs as fd  # noqa: E402
import ref_harness  # noqa: E402


def probe_tree(L, anchors):
    tiny = 2.0 ** -24
    probes = [(i, j, k) for i in anchors for j in range(L) for k in range(j + 1, L) if i != j and i != k]
    M = torch.zeros((len(probes) + 1, L), dtype=torch.float32)
    Mn = M.numpy()
    for t, (i, j, k) in enumerate(probes):
        Mn[t, i] = 1.0
        Mn[t, j] = tiny
        Mn[t, k] = tiny
    Mn[-1] = 1.0
    y = M.matmul(M[-1]).numpy()[:-1]
    res = {p: bool(y[t] > 1.0) for t, p in enumerate(probes)}
    for i in anchors:
        parent = list(range(L))

        def find(a):
            while parent[a] != a:
                a = parent[a]
            return a

        for j in range(L):
            for k in range(j + 1, L):
                if i not in (j, k) and res[(i, j, k)]:
                    parent[find(j)] = find(k)
        groups = {}
        for j in range(L):
            if j != i:
                groups.setdefault(find(j), []).append(j)
        print(f"  L {L} anchor {i}: columns that meet each other before they meet the anchor: "
              f"{sorted(groups.values(), key=lambda v: v[0])}")


def main():
    print("torch", torch.__version__, "|", [l.strip() for l in torch.__config__.show().splitlines() if "Math Kernel" in l or "CPU capability" in l])
    print("1. summation tree of matrix.matmul(vector):")
    probe_tree(32, [0, 1, 16, 17, 31])
    probe_tree(40, [0, 17, 33, 39])
    _, cl, _ = ref_harness.load_reference()
    print("2. oracle/cluster_scan.c in reference order against torch, bits that differ:")
    rng = np.random.RandomState(7)
    for threads in (1, 8):
        torch.set_num_threads(threads)
        tot_d = tot_n = cnt = 0
        for L in list(range(1, 70)) + [80, 100, 128, 129, 132, 200, 257]:
            N = 20000 if L <= 69 else 5000
            raw = (rng.standard_normal((N, L)) * rng.uniform(0.1, 3, (N, 1))).astype(np.float32)
            ref_norm = cl._normalize(torch.from_numpy(raw.copy())).numpy()
            co.set_order(1)
            ours = co.normalize(raw.copy())
            tot_n += int((ours.view(np.uint32) != ref_norm.view(np.uint32)).sum())
            t = torch.from_numpy(ref_norm)
            for idx in (0, N // 3, N - 1):
                want = cl._calc_distances(t, idx).numpy()
                got = co.scan(ref_norm, np.ones(N, np.float32), None, idx)["dist"]
                tot_d += int((got.view(np.uint32) != want.view(np.uint32)).sum())
                cnt += N
        print(f"  threads {threads}: latent widths 1..69, 80..257: normalised elements differing {tot_n}, distances differing {tot_d} of {cnt}")
    co.set_order(co.DEFAULT_ORDER)


if __name__ == "__main__":
    main()
