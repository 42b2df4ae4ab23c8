#!/usr/bin/env python
"""Diagnostic (not a test): where the time of a many-medoid scan pass goes -- the pass with parts switched off (library option
scan.debug, WRONG results): 1 = no pair of interest (loads + matrix pipe only), 8 = queued pairs dropped (hit extraction but no
drain), 2 = no histogram, 4 = no flush.   python tools/gpu/gpu_scan_dbg.py [n] [k] [out.txt]

Medoid sets: `random` rows, and `neighbours` = k rows of one genome (what the generator's speculation puts into one pass)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import cluster as vc, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 620_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = sys.argv[3] if len(sys.argv) > 3 else None
L = 32
lat, labels = synth.blob_latent(n, L, 0.3, seed=1)
lens = synth.lengths(n, 1).astype(np.float32)
rng = np.random.RandomState(0)
g = labels[rng.randint(n)]
same = np.flatnonzero(labels == g)
sets = {"random": rng.choice(n, k, replace=False), "neighbours": same[:k] if len(same) >= k else rng.choice(n, k, replace=False)}
lines = []
for dbg in [int(x) for x in os.environ.get('SCAN_DBG_LIST', '0,1,8,2,4,12,13').split(',')]:
    os.environ["VAMBHIP_SCAN_DBG"] = str(dbg)
    b = vc.HipScanBackend(lat, lens, False, None)
    for name, med in sets.items():
        b.scan_raw(med)
        b.set_timing(True)
        b.kernel_ms = 0.0
        reps = 20
        for _ in range(reps):
            b.scan_raw(med)
        kus = b.kernel_ms / reps * 1e3
        b.set_timing(False)
        l = f"n={n} k={k} medoids={name:10s} scan.debug={dbg:2d}: kernel {kus:7.1f} us"
        print(l, flush=True)
        lines.append(l)
    b.close()
os.environ.pop("VAMBHIP_SCAN_DBG", None)
if out:
    open(out, "w").write("\n".join(lines) + "\n")
