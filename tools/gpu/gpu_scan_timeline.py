#!/usr/bin/env python
"""Diagnostic (not a test; needs VAMBHIP_LIB_PATH = a library built by `python tools/build_variant.py timing -DVAMBHIP_TIMING_EXPERIMENTS`):
where the time of ONE scan pass goes -- constant-clock stamps written by the scan / publish kernels (vh_debug_scan_timeline) and the
host's own clock around launch and flag.   python tools/gpu/gpu_scan_timeline.py n L k [reps] [sigma]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vamb_amd import _lib, cluster as vc, synth  # noqa: E402

n, L, k = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
lat, _ = synth.blob_latent(n, L, sigma, seed=1)
b = vc.HipScanBackend(lat, synth.lengths(n, 1).astype(np.float32), False, None)
med = np.random.RandomState(0).choice(n, k, replace=False)
ROWS = 4096
stamps = np.zeros((ROWS, 8), np.uint64)
host = np.zeros(4, np.float64)
nrows = ctypes.c_int(0)
acc = []
hosts = []
b.set_timing(True)
kms = []
for r in range(reps + 3):
    b.kernel_ms = 0.0
    b.scan_raw(med)
    kms.append(b.kernel_ms * 1e3)
    _lib.check(b.lib.vh_debug_scan_timeline(b.h, _lib.ptr(stamps), ROWS, ctypes.byref(nrows), _lib.ptr(host)))
    if r < 3:
        continue
    s = stamps.astype(np.int64)
    wg = s[:ROWS - 1]
    ran = wg[:, 0] > 0
    t0 = wg[ran, 0].min()
    ph = {}
    names = ["entry", "prologue done", "first block evaluated", "row loop done", "drain done", "flush begun", "flush retired"]
    for i, nm in enumerate(names):
        col = wg[ran, i]
        col = col[col > 0]
        if len(col):
            ph[nm] = ((col.min() - t0) / 100.0, (np.median(col) - t0) / 100.0, (col.max() - t0) / 100.0)
    pub = s[ROWS - 1]
    ph["publish entry"] = ((pub[0] - t0) / 100.0,) * 3
    ph["publish copies added"] = ((pub[1] - t0) / 100.0,) * 3
    ph["publish done"] = ((pub[2] - t0) / 100.0,) * 3      # (two-step publish: the pass flag; the histograms follow)
    if pub[3] > 0:
        ph["publish histograms done"] = ((pub[3] - t0) / 100.0,) * 3
    acc.append((int(ran.sum()), ph))
    hosts.append(host.copy())
b.set_timing(False)
print(f"n={n} L={L} k={k} sigma={sigma}: {acc[0][0]} workgroups; event-timed scan kernel {np.median(kms[3:]):.1f} us (median of {reps})")
print("  device stamps, us after the first workgroup's entry (min / median / max over workgroups; median over passes):")
for nm in acc[0][1]:
    v = np.median(np.array([a[1][nm] for a in acc if nm in a[1]]), axis=0)
    print(f"    {nm:24s} {v[0]:8.2f} {v[1]:8.2f} {v[2]:8.2f}")
h = np.median(np.array(hosts), axis=0)
print(f"  host clock, us after scan_core's entry: scan launched {h[1]:.2f}, publish launched {h[2]:.2f}, flag seen {h[3]:.2f}  (event timing on: the stream carries two more event records)")
# the same passes without event timing: the wall time the state machine sees
import time
t0 = time.perf_counter()
for _ in range(50):
    b.scan_raw(med)
print(f"  pass wall time without event timing: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
_lib.check(b.lib.vh_debug_scan_timeline(b.h, _lib.ptr(stamps), ROWS, ctypes.byref(nrows), _lib.ptr(host)))
print(f"  host clock of the last such pass: scan launched {host[1]:.2f}, publish launched {host[2]:.2f}, flag seen {host[3]:.2f}")
b.close()
