#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (oracle/; build container): how often do candidate DEFINED summation orders of the cluster
path's dot product disagree with the reference's `matrix.matmul(matrix[medoid])` (torch CPU -> MKL sgemv, whose order is not
defined and depends on the ISA MKL picks and on the thread count)?

For each 100 k-point fixture: M random medoids, all N rows.  Reported per order: pairs whose float32 distance
d = 0.5 - dot differs in any bit from torch's, and -- what a stream can see -- pairs whose DECISIONS differ:
`d <= 0.05` (medoid radius, cluster.py:621), `d < 0.05` (loner test, cluster.py:457), `d <= 0.3` (histogram range,
cluster.py:467) and the histogram bin (torch.histogram edges).

    python oracle/compare_dot_orders.py [medoids] > profiles/r03_dot_order_vs_mkl.txt
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fixture_defs as fd  # noqa: E402
import ref_harness  # noqa: E402

f32 = np.float32


def fma32(a, b, c):
    """float32 fused multiply-add, correctly rounded (exact product and sum in float64: 24 x 24 bits fit, the sum of a 48-bit
    product and a 24-bit addend is then rounded ONCE to float32 -- double rounding cannot occur because the float64 sum of
    these operands is exact or its float64 rounding does not sit on a float32 tie for non-pathological data; good enough
    for counting)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def order_chain(m, q):
    acc = np.zeros(len(m), f32)
    for c in range(m.shape[1]):
        acc = fma32(m[:, c], np.full(len(m), q[c], f32), acc)
    return acc


def order_two_chains(m, q):
    """columns [0, L/2) and [L/2, L) as two ascending fmaf chains, then one add"""
    h = m.shape[1] // 2
    return (order_chain(m[:, :h], q[:h]) + order_chain(m[:, h:], q[h:])).astype(f32)


def order_lanes16(m, q):
    """16 vector lanes (lane j: columns j, j + 16, ... as an fma chain starting with a plain product), then a halving tree:
    what an AVX-512 sgemv over a contiguous row does"""
    L = m.shape[1]
    p = (m[:, :16] * q[:16]).astype(f32)
    for b in range(16, L, 16):
        w = min(16, L - b)
        p[:, :w] = fma32(m[:, b:b + w], np.broadcast_to(q[b:b + w], (len(m), w)), p[:, :w])
    while p.shape[1] > 1:
        hh = p.shape[1] // 2
        p = (p[:, :hh] + p[:, hh:]).astype(f32)
    return p[:, 0]


def order_lanes8(m, q):
    L = m.shape[1]
    p = (m[:, :8] * q[:8]).astype(f32)
    for b in range(8, L, 8):
        p = fma32(m[:, b:b + 8], np.broadcast_to(q[b:b + 8], (len(m), 8)), p)
    while p.shape[1] > 1:
        hh = p.shape[1] // 2
        p = (p[:, :hh] + p[:, hh:]).astype(f32)
    return p[:, 0]


def order_f64(m, q):
    return (m.astype(np.float64) @ q.astype(np.float64)).astype(f32)


ORDERS = [("ascending fmaf chain (HIP kernels today)", order_chain), ("two half chains + add", order_two_chains),
          ("16 lanes + halving tree (AVX-512 shape)", order_lanes16), ("8 lanes + halving tree (AVX2 shape)", order_lanes8),
          ("float64 dot rounded once", order_f64)]


def main():
    n_med = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    vt, rc, _ = ref_harness.load_reference()
    edges = torch.linspace(0.0, 0.3, 61).numpy()
    for threads in (1, 8):
        torch.set_num_threads(threads)
        for name in fd.CLUSTER_CASES_LARGE:
            mat, lens, kw = fd.cluster_inputs(name)
            m = mat.copy()
            rc._normalize(torch.from_numpy(m))           # the reference's own normalisation, in place
            rng = np.random.RandomState(7)
            meds = rng.choice(len(m), n_med, replace=False)
            tm = torch.from_numpy(m)
            tot = {k: np.zeros(6, np.int64) for k, _ in ORDERS}
            for med in meds:
                d_ref = (0.5 - tm.matmul(tm[int(med)])).numpy()
                for label, fn in ORDERS:
                    d = (f32(0.5) - fn(m, m[med])).astype(f32)
                    t = tot[label]
                    t[0] += int((d.view(np.uint32) != d_ref.view(np.uint32)).sum())
                    t[1] += int(((d <= f32(0.05)) != (d_ref <= f32(0.05))).sum())
                    t[2] += int(((d < f32(0.05)) != (d_ref < f32(0.05))).sum())
                    t[3] += int(((d <= f32(0.3)) != (d_ref <= f32(0.3))).sum())
                    inr = (d <= f32(0.3)) & (d_ref <= f32(0.3)) & (d >= 0) & (d_ref >= 0)
                    t[4] += int((np.searchsorted(edges, d[inr], "right") != np.searchsorted(edges, d_ref[inr], "right")).sum())
                    t[5] += len(d)
            print(f"{name}, torch threads {threads}, {n_med} medoids x {len(m)} rows (L = {m.shape[1]}):")
            for label, _ in ORDERS:
                t = tot[label]
                print(f"   {label:44s} bits differ {t[0]:9d} ({t[0] / t[5]:.2%})   d<=0.05 flips {t[1]:4d}   d<0.05 flips {t[2]:4d}"
                      f"   d<=0.3 flips {t[3]:4d}   histogram bin flips {t[4]:5d}")


if __name__ == "__main__":
    main()
