import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from vamb_amd import cluster as vc, synth
for (n, L) in [(100_000, 32), (250_000, 32), (500_000, 32)]:
    lat, _ = synth.blob_latent(n, L, 0.3, seed=1)
    lens = synth.lengths(n, 1)
    b = vc.HipScanBackend(lat, lens.astype(np.float32), False, None)
    rng = np.random.RandomState(0)
    for k in (4, 8, 9, 12, 16, 24, 32):
        med = rng.choice(n, k, replace=False)
        b.scan_raw(med)
        b.set_timing(True); b.kernel_ms = 0.0
        for _ in range(20): b.scan_raw(med)
        kms = b.kernel_ms / 20
        b.set_timing(False)
        t0 = time.perf_counter()
        for _ in range(100): b.scan_raw(med)
        wall = (time.perf_counter() - t0) / 100
        print(f"mfma={os.environ.get('VAMBHIP_SCAN_MFMA','1')} n={n} k={k:2d}: kernel {kms*1e3:6.1f} us  pass wall {wall*1e6:6.1f} us", flush=True)
    b.close()
