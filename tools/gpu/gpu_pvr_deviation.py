#!/usr/bin/env python
"""Diagnostic: how far the REPORTED observed_pvr of the library (exact histogram sums) is from the reference's (torch.histogram's
float32 bin sums) on the committed golden streams, and whether round(pvr, 2) -- what vamb/__main__.py:1368-1369 writes into
*_metadata.tsv -- ever differs.   python tools/gpu/gpu_pvr_deviation.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fixture_defs as fd  # noqa: E402
from vamb_amd import _lib, cluster as vc  # noqa: E402
_lib.require_gpu()
names = list(fd.CLUSTER_CASES_LARGE) + [n for n in fd.CLUSTER_CASES if n not in fd.CLUSTER_CASES_LARGE]
worst = 0.0
for name in names:
    try:
        mat, lens, kw = fd.cluster_inputs(name)
        golden = fd.load("cluster_" + name)
    except Exception as e:  # fixtures without a golden stream
        print(name, "skipped:", e); continue
    got = fd.pack_stream(list(vc.ClusterGenerator(mat.copy(), lens, **kw)))
    a, b = got["observed_pvr"], golden["observed_pvr"]
    if a.shape != b.shape:
        print(name, "stream lengths differ", a.shape, b.shape); continue
    m = ~np.isnan(a) & ~np.isnan(b)
    d = np.abs(a[m] - b[m])
    nz = b[m] != 0
    rel = d[nz] / np.abs(b[m][nz]) if nz.any() else np.zeros(1)
    if (~nz).any():
        assert np.all(d[~nz] == 0), "a ratio the reference reports as 0 is not 0 here"
    r2 = int(np.sum(np.round(a[m], 2) != np.round(b[m], 2)))
    worst = max(worst, float(rel.max()))
    print(f"{name}: {int(m.sum())} reported ratios, max rel deviation {rel.max():.3e}, max abs {d.max() if d.size else 0.0:.3e}, "
          f"bit-identical {int(np.sum(a[m] == b[m]))}, round(.,2) differs in {r2}", flush=True)
print(f"worst relative deviation over all fixtures: {worst:.3e}")
