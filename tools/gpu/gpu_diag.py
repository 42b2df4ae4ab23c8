#!/usr/bin/env python
"""GPU-box diagnostic (not a test): prints error magnitudes and micro-benchmarks so one gpurun call
tells us what is wrong and how fast the kernels are.  Writes gpurun_out/diag.json."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from vamb_amd import _lib, cluster as vc, synth  # noqa: E402

out = {}
lib = _lib.load()
print("devices:", _lib.device_count(), lib.vh_version())


def gemm_bench():
    res = []
    rng = np.random.RandomState(0)
    shapes = []
    for tile in (0, 1, 2, 3):
        shapes += [(tile, 1, 1, 4096, 512, 160, 1), (tile, 1, 1, 4096, 512, 512, 1), (tile, 1, 1, 4096, 160, 512, 1),
                   (tile, 1, 0, 4096, 512, 512, 1), (tile, 1, 0, 4096, 512, 160, 1),
                   (tile, 0, 0, 512, 512, 4096, 8), (tile, 0, 0, 512, 512, 4096, 16), (tile, 0, 0, 512, 160, 4096, 16),
                   (tile, 1, 1, 8192, 512, 1120, 1)]
    shapes += [(2, 1, 1, 4096, 32, 512, 1), (2, 1, 1, 4096, 32, 512, 8), (2, 1, 0, 4096, 32, 512, 8)]
    for (tile, a_kc, b_kc, M, N, K, splits) in shapes:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        Ad = np.ascontiguousarray(A if a_kc else A.T)
        Bd = np.ascontiguousarray(B if b_kc else B.T)
        C = np.zeros((M, N), np.float32)
        ms = ctypes.c_float()
        st = lib.vh_debug_gemm(tile, a_kc, b_kc, _lib.ptr(Ad), _lib.ptr(Bd), None, _lib.ptr(C), M, N, K, splits,
                               ctypes.byref(ms))
        if st != 0:
            res.append(dict(tile=tile, layout=(a_kc, b_kc), shape=(M, N, K), error=lib.vh_last_error().decode()))
            continue
        want = A.astype(np.float64) @ B.astype(np.float64).T
        err = float(np.abs(C - want).max() / np.abs(want).max())
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        res.append(dict(tile=tile, layout=(a_kc, b_kc), shape=(M, N, K), splits=splits, ms=ms.value, tflops=tf,
                        relerr=err))
        print("gemm", res[-1])
    return res


def scan_bench():
    res = []
    for (n, L) in [(200_000, 32), (2_000_000, 32), (2_000_000, 64)]:
        lat, _ = synth.blob_latent(n, L, 0.08, seed=1)
        lens = synth.lengths(n, 1)
        b = vc.HipScanBackend(lat, lens.astype(np.float32), False, None)
        b.set_timing(True)
        rng = np.random.RandomState(0)
        for k in (1, 2, 4, 8, 12, 16, 24, 25, 32):
            meds = [int(x) for x in rng.choice(n, size=k, replace=False)]
            b.scan(meds)  # warm
            b.kernel_ms = 0.0
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                b.scan(meds)
            wall = (time.perf_counter() - t0) / reps * 1e3
            kms = b.kernel_ms / reps
            bytes_ = n * (4 * ((L + 3) // 4 * 4) + 5)
            res.append(dict(n=n, L=L, k=k, kernel_ms=kms, wall_ms=wall, GBps=bytes_ / (kms * 1e-3) / 1e9))
            print("scan", res[-1])
            if k in (1, 16):
                # the same pass with explicit query vectors (the row-sharded path): isolates the in-kernel gather
                q = b.get_rows(meds)
                b.scan_raw(meds, queries=q)
                b.kernel_ms = 0.0
                for _ in range(reps):
                    b.scan_raw(meds, queries=q)
                res.append(dict(n=n, L=L, k=k, mode="explicit_queries", kernel_ms=b.kernel_ms / reps))
                print("scan", res[-1])
        b.kernel_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(5):
            b.select(meds[0], 0.05, remove=False)
        res.append(dict(n=n, L=L, select_kernel_ms=b.kernel_ms / 5, select_wall_ms=(time.perf_counter() - t0) / 5 * 1e3))
        print("select", res[-1])
        b.close()
    return res


def cluster_e2e():
    res = []
    for n in (50_000, 200_000):
        lat, _ = synth.blob_latent(n, 32, 0.08, seed=1)
        lens = synth.lengths(n, 1)
        t0 = time.perf_counter()
        g = vc.ClusterGenerator(lat, lens, destroy=True, rng_seed=1)
        g._backend.set_timing(True)
        nc = sum(1 for _ in g)
        dt = time.perf_counter() - t0
        b = g._backend
        res.append(dict(n=n, clusters=nc, seconds=dt, passes=b.scan_passes, medoids=b.scan_medoids,
                        rows_streamed=b.rows_streamed, kernel_ms=b.kernel_ms))
        print("cluster", res[-1])
    return res


def vae_bench():
    from vamb_amd import encode as ve
    res = []
    for (n, S, bs) in [(200_000, 50, 4096)]:
        ab, tnf, lens, _ = synth.features(n, S, seed=1)
        dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
        vae = ve.VAE(S, seed=1)
        t0 = time.perf_counter(); vae.trainmodel(dl, nepochs=1, batchsteps=None); t_first = time.perf_counter() - t0
        _lib.check(lib.vh_vae_set_probe(vae._h, 1, 1))
        t0 = time.perf_counter(); vae.trainmodel(dl, nepochs=5, batchsteps=None); t5 = time.perf_counter() - t0
        ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.vh_vae_probe_result(vae._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
        t0 = time.perf_counter(); lat = vae.encode(dl); te = time.perf_counter() - t0
        res.append(dict(n=n, S=S, bs=bs, first_epoch_s=t_first, epoch_s=t5 / 5, contigs_per_s_epoch=n / (t5 / 5),
                        encode_s=te, probe_ms_avg=ms.value / max(1, nl.value), probe_launches=nl.value,
                        probe_tflops=fl.value / (ms.value / max(1, nl.value) * 1e-3) / 1e12 if nl.value else None,
                        loss=vae.last_epoch_losses))
        print("vae", res[-1])
    return res


which = sys.argv[1:] or ["gemm", "scan", "cluster", "vae"]
for name, fn in (("gemm", gemm_bench), ("scan", scan_bench), ("cluster", cluster_e2e), ("vae", vae_bench)):
    if name not in which:
        continue
    try:
        out[name] = fn()
    except Exception as e:  # keep going: we want as much information per GPU call as possible
        import traceback
        traceback.print_exc()
        out[name] = dict(error=repr(e))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1)
