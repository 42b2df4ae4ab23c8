#!/usr/bin/env python
"""bench.py -- whole-job throughput of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--epochs E] [--contigs N] [--samples S]

One "step" = one full pass of the hot path over a synthetic dataset resident in HBM:
    VAE.trainmodel (E epochs, fixed batch) -> VAE.encode -> list(ClusterGenerator(latent))
i.e. exactly what `vamb bin default` runs between loading the abundance/TNF matrices and writing
cluster files (reference vamb/__main__.py:1451-1488).  The default workload is BASELINE.json
configs[1] ("C1"): 200k contigs x 50 samples (D = 154), 512-512 hidden, 32-d latent, batch 4096, fp32,
with the reference CLI's default epoch count (-e 300, __main__.py:2412).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : HIP-event timing (the kernels' own begin/end timestamps) of the encoder layer-1 GEMM
                 (M=batch, K=512, N=512 -- the FLOP-dominant encoder GEMM) over the timed region
  cpu_baseline : the CPU oracle ("port") timed on a bounded sample of the same workload
and extra per-stage fields (epoch_ms, encode_ms, cluster_ms, scan GB/s).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- every rank holds its own
shard of `--contigs` contigs; training is data-parallel (RCCL all-reduce of the flat gradient on the
library's stream, see DESIGN.md "multi-GPU"), encode and the cluster sweep are shard-local.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HIDDEN = 512
NTNF = 103
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md chip table (fp32-input MFMA == fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same table); only used with VAMBHIP_PRECISION=bf16
PEAK_HBM_GBPS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--epochs", type=int, default=300, help="training epochs per step (reference CLI default 300)")
    p.add_argument("--contigs", type=int, default=200_000, help="contigs per GPU")
    p.add_argument("--samples", type=int, default=50)
    p.add_argument("--batch", type=int, default=4096)
    p.add_argument("--latent", type=int, default=32)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=20_000)
    p.add_argument("--force-dist", action="store_true",
                   help="take the multi-GPU code path (process group + RCCL communicator) even with one rank")
    return p.parse_args()


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local, world


def run_step(ve, vc, lib, _lib, dl, lens, args, seed, comm=None, probe_layer=1, time_scans=False):
    """One pass of the hot path.  Returns per-stage seconds and counters."""
    t0 = time.perf_counter()
    vae = ve.VAE(args.samples, nlatent=args.latent, seed=seed)
    if comm is not None:
        vae.attach_communicator(comm)
    _lib.check(lib.vh_vae_set_probe(vae._h, 1, probe_layer))
    vae._ensure_dataset(dl)
    t1 = time.perf_counter()
    vae.trainmodel(dl, nepochs=args.epochs, batchsteps=None)
    t2 = time.perf_counter()
    latent = vae.encode(dl)
    t3 = time.perf_counter()
    gen = vc.ClusterGenerator(latent, lens, destroy=True, rng_seed=seed)
    # HIP-event timing of every scan / select kernel costs a stream synchronisation per pass: it is switched on
    # in the warm-up steps only (cluster_scan statistics), the timed steps run the sweep as a user would
    gen._backend.set_timing(time_scans)
    n_clusters = 0
    n_points = 0
    for c in gen:
        n_clusters += 1
        n_points += len(c.members)
    t4 = time.perf_counter()
    assert n_points == len(lens)
    ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(lib.vh_vae_probe_result(vae._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
    b = gen._backend
    L4 = (args.latent + 3) // 4 * 4
    out = dict(setup_s=t1 - t0, train_s=t2 - t1, encode_s=t3 - t2, cluster_s=t4 - t3, total_s=t4 - t0,
               clusters=n_clusters, probe_ms=ms.value, probe_launches=nl.value, probe_flops=fl.value,
               scan_passes=b.scan_passes, scan_medoids=b.scan_medoids, scan_kernel_ms=b.kernel_ms,
               scan_bytes=b.rows_streamed * (4 * L4 + 5), loss=vae.last_epoch_losses["loss"], latent=latent)
    b.close()
    return out


def pmc_traffic():
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC pass (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_roofline_kernel.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def layer0_roofline(warm):
    ms = sum(r["probe_ms"] for r in warm)
    n = sum(r["probe_launches"] for r in warm)
    if not n:
        return None
    flops = warm[-1]["probe_flops"]
    ach = flops / (ms / n * 1e-3) / 1e12
    return {"kernel": "gemm_f32_kernel<64,64,EPI_HIDDEN_TRAIN> (encoder layer 0: M=batch, K=D, N=512), timed in the warm-up steps",
            "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
            "avg_launch_ms": ms / n, "launches": n, "flops_per_launch": flops}


def cpu_baseline(args, latent, lens):
    """The oracle (numpy VAE restatement + C cluster restatement) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cluster_oracle as co
    import vae_oracle as vo
    from vamb_amd import encode as ve, synth

    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    n = min(args.cpu_sample, args.contigs)
    ab, tnf, ln, _ = synth.features(n, args.samples, seed=101)
    dl = ve.make_dataloader(ab, tnf, ln, batchsize=args.batch, destroy=True)
    d, t, a, w = (x.numpy() for x in dl.dataset.tensors)
    st = vo.init_state(args.samples, [HIDDEN, HIDDEN], args.latent, 1)
    m = vo.OracleVAE(args.samples, [HIDDEN, HIDDEN], args.latent, None, 200.0, 0.2, state=st, dtype=np.float32)
    rng = np.random.RandomState(0)
    bs = min(args.batch, n)
    nb = max(1, n // bs)
    cpu_epochs = 3
    t0 = time.perf_counter()
    for _ in range(cpu_epochs):
        perm = rng.permutation(n)
        for b in range(nb):
            rows = perm[b * bs:(b + 1) * bs]
            masks = [rng.random_sample((len(rows), HIDDEN)) >= 0.2 for _ in range(4)]
            eps = rng.standard_normal((len(rows), args.latent)).astype(np.float32)
            m.train_step(d[rows], t[rows], a[rows], w[rows], eps, masks)
    t_epoch = (time.perf_counter() - t0) / cpu_epochs
    t0 = time.perf_counter()
    m.encode(d, t, a)
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    nclu = sum(1 for _ in co.OracleClusterGenerator(np.ascontiguousarray(latent[:n]), lens[:n], rng_seed=1))
    t_clu = time.perf_counter() - t0
    total = args.epochs * t_epoch + t_enc + t_clu
    return dict(value=n / total, unit="contigs/s", cores=int(cores), kind="port",
                sample=(f"{n} contigs x {args.samples} samples: {cpu_epochs} oracle epochs timed "
                        f"({t_epoch:.3f} s/epoch, numpy fp32 BLAS) extrapolated to {args.epochs}, + encode "
                        f"{t_enc:.3f} s + full cluster sweep of the first {n} GPU latents {t_clu:.3f} s "
                        f"({nclu} clusters, scalar C)"),
                epoch_s=t_epoch, encode_s=t_enc, cluster_s=t_clu)


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  gloo and RCCL print banners there, RCCL through C stdio that is
    only flushed at exit (i.e. AFTER the result line).  So file descriptor 1 is pointed at stderr for the whole
    run and the result line is written to a private duplicate of the original stdout."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def emit_result(line: dict):
    os.write(_RESULT_FD, (json.dumps(line) + "\n").encode())


def main():
    claim_stdout()
    args = parse()
    rank, local, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    from vamb_amd import _lib, cluster as vc, encode as ve, synth

    lib = _lib.load()
    _lib.require_gpu()
    _lib.check(lib.vh_set_device(local))
    comm = None
    dist = None
    if world > 1 or args.force_dist:
        # Control plane: gloo (bootstrap, barriers, the max-over-ranks of the timing).  Data plane: RCCL
        # called inside libvambhip on its own HIP stream (vamb_amd/parallel.py).  torch.cuda is never
        # touched: PyTorch-ROCm bundles a private HIP runtime that must not be mixed with the library's.
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="gloo")
        from vamb_amd import parallel

        comm = parallel.Communicator.from_torch_distributed(dist)

    # synthetic inputs of the named shape (per-rank shard under weak scaling), normalised on the host
    # exactly as `vamb bin default` does, then uploaded once: resident in HBM before the clock starts
    ab, tnf, lens, _ = synth.features(args.contigs, args.samples, seed=1 + rank)
    # under data parallelism the loader's batch size is the ALL-RANK batch: --batch rows per GPU
    dl = ve.make_dataloader(ab, tnf, lens, batchsize=args.batch * world, destroy=True)

    def barrier():
        _lib.check(lib.vh_device_synchronize())
        if dist is not None:
            dist.barrier()
            _lib.check(lib.vh_device_synchronize())

    warm = []
    for i in range(args.warmup):
        # the warm-up steps time the (much smaller) layer-0 GEMM instead; reported as roofline_layer0
        warm.append(run_step(ve, vc, lib, _lib, dl, lens, args, seed=1000 + i, comm=comm, probe_layer=0,
                             time_scans=True))
    barrier()
    t0 = time.perf_counter()
    results = []
    for i in range(args.steps):
        results.append(run_step(ve, vc, lib, _lib, dl, lens, args, seed=i, comm=comm, time_scans=args.warmup == 0))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        K = max(1, args.steps)
        total_contigs = args.contigs * world * K
        D = args.samples + NTNF + 1
        probe_ms = sum(r["probe_ms"] for r in results)
        probe_n = sum(r["probe_launches"] for r in results)
        flops = results[-1]["probe_flops"] if results else 0.0
        avg_ms = probe_ms / probe_n if probe_n else float("nan")
        achieved = flops / (avg_ms * 1e-3) / 1e12 if probe_n else float("nan")
        bf16 = ve.get_compute_dtype() == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        shape = (args.contigs, args.samples, args.batch, args.latent)
        cfg_name = {(200_000, 50, 4096, 32): "C1", (2_000_000, 200, 8192, 32): "C2",
                    (2_000_000, 1000, 8192, 32): "C3", (10_000_000, 1000, 8192, 64): "C4"}.get(shape, "custom")
        scan_src = warm if warm else results          # the steps that ran with scan-kernel timing on
        scan_ms = sum(r["scan_kernel_ms"] for r in scan_src)
        scan_bytes = sum(r["scan_bytes"] for r in scan_src)
        line = {
            "metric": "contigs/sec through VAE-train+encode+cluster; VAE epoch step time",
            "value": total_contigs / elapsed,
            "unit": "contigs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{cfg_name}: {args.contigs} contigs x {args.samples} samples per GPU (D={D}), hidden 512-512, "
                             f"latent {args.latent}, batch {args.batch}, {'bf16 MFMA / fp32 accumulate' if bf16 else 'fp32 MFMA'}; "
                             f"{args.epochs} train epochs "
                             f"(reference CLI default) + encode + full cluster sweep per step"),
                "contigs_per_gpu": args.contigs, "samples": args.samples, "batch": args.batch,
                "epochs": args.epochs, "parallelism": f"dp{world}" if world > 1 else "single",
            },
            "roofline": {
                "kernel": ("gemm_f32_kernel<64,64,2x2 waves,EPI_HIDDEN_TRAIN,XF_BN> (encoder layer 1, the FLOP-dominant "
                           "encoder GEMM: M=batch, K=512, N=512; BatchNorm of layer 0 applied on load, "
                           "bias+leaky-relu+dropout+BN batch sums in the epilogue)"),
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak if probe_n else None, "traffic": pmc_traffic() if cfg_name == "C1" else None,
                "avg_launch_ms": avg_ms, "launches": probe_n, "flops_per_launch": flops,
                "timing": "kernel begin/end timestamps (hipExtLaunchKernelGGL start/stop events) of every launch in the timed steps",
            },
            "roofline_layer0": layer0_roofline(warm),
            "epoch_ms": np.mean([r["train_s"] for r in results]) / args.epochs * 1e3,
            "train_contigs_per_s_per_epoch": args.contigs * world / (np.mean([r["train_s"] for r in results]) / args.epochs),
            "encode_ms": np.mean([r["encode_s"] for r in results]) * 1e3,
            "cluster_ms": np.mean([r["cluster_s"] for r in results]) * 1e3,
            "clusters_per_step": int(np.mean([r["clusters"] for r in results])),
            "cluster_scan": {
                "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBPS,
                "achieved": scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms else None,
                "frac": scan_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if scan_ms else None,
                "passes": sum(r["scan_passes"] for r in scan_src), "medoids": sum(r["scan_medoids"] for r in scan_src),
                "kernel_ms_total": scan_ms,
                "measured_in": "warm-up steps" if warm else "timed steps",
            },
            "final_loss": results[-1]["loss"] if results else None,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, results[-1]["latent"], lens)
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        emit_result(line)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
