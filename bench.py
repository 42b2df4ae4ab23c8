#!/usr/bin/env python
"""bench.py -- whole-job throughput of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C1|C3] [--epochs E] [--scaling weak|strong]

One "step" = one full pass of the hot path over a synthetic dataset resident in HBM:
    VAE.trainmodel (E epochs, fixed batch) -> VAE.encode -> list(ClusterGenerator(latent))
i.e. what `vamb bin default` runs between loading the abundance / TNF matrices and writing the cluster files
(reference vamb/__main__.py:1065-1107, 1254-1404).  The default workload is BASELINE.json configs[2] ("C2"), the
largest single-GPU configuration: 2 M contigs x 200 samples (D = 304), 512-512 hidden, 32-d latent, batch 8192,
bf16 MFMA with fp32 accumulation, the reference CLI's default epoch count (-e 300, __main__.py:2412).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline        : kernel begin/end timestamps of the encoder's FIRST-layer GEMM (M = batch, K = D, N = 512 -- the
                    GEMM BASELINE's >= 50 % target names) over the timed steps, against the dtype's MFMA peak
  roofline_hidden : the same for a 512x512 hidden layer (the GEMM shape that dominates the step's time), warm-up steps
  cluster_scan    : algorithmic bytes / kernel time of the scan + select passes (HBM bound), warm-up steps
  c1, c3_shape    : the WHOLE job (300 epochs + encode + sweep) at configs[1] (200 k x 50, fp32) and at the C3 shape (2 M x 1000,
                    D = 1104, the shape north_star's 1-GPU target is quoted on) on this GPU, each with the headline's protocol:
                    one full warm-up job, then --leg-steps (3) timed jobs between two device synchronisations
  cluster_order   : which arithmetic the sweeps ran in (library option scan.reference_order; default = the reference's order)
  cpu_baseline    : the CPU oracle ("port") timed on a bounded sample, with its calibration against the real
                    reference measured in the build container (oracle/cpu_calibration.json)

N > 1: `python bench.py --gpus N` launches N ranks itself (one process per GPU; the driver's
`python -m torch.distributed.run ... bench.py --gpus N` form works too).  Default (--scaling strong, --config C3 = BASELINE's
8-GPU configuration): ONE dataset of 2 M contigs x 1000 samples row-sharded over the ranks -- data-parallel training with
synchronised BatchNorm (RCCL all-reduce of the flat gradient and of the BatchNorm sums inside the library), shard-local encode,
and the cluster sweep through the NATIVE sharded state machine (vh_gen_create_sharded: every rank scans its shard, one
all-gather of exact integer accumulators per pass); on one GPU the same workload is the `c3_shape` object of the N = 1 line.
--scaling weak: every rank holds its own dataset of `--contigs` contigs and sweeps it locally (N independent jobs).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.perf_counter()

HIDDEN = 512
NTNF = 103
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md chip table (fp32-input MFMA == fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same table)
PEAK_HBM_GBPS = 8000.0

CONFIGS = {  # BASELINE.json configs: (contigs, samples, batch, latent, dtype)
    "C1": (200_000, 50, 4096, 32, "fp32"),
    "C2": (2_000_000, 200, 8192, 32, "bf16"),
    "C3": (2_000_000, 1000, 8192, 32, "bf16"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", choices=sorted(CONFIGS), default=None,
                   help="default: C2 on one GPU (the largest single-GPU configuration); C3 -- BASELINE's 8-GPU configuration, "
                        "ONE 2 M x 1000 dataset row-sharded over the ranks -- on more than one")
    p.add_argument("--epochs", type=int, default=300, help="training epochs per step (reference CLI default 300)")
    p.add_argument("--contigs", type=int, default=None, help="contigs per GPU (weak) / in total (strong)")
    p.add_argument("--samples", type=int, default=None)
    p.add_argument("--batch", type=int, default=None, help="rows per GPU and optimisation step")
    p.add_argument("--latent", type=int, default=None)
    p.add_argument("--dtype", choices=["fp32", "bf16"], default=None)
    p.add_argument("--scaling", choices=["weak", "strong"], default=None,
                   help="N > 1: strong (default) = ONE dataset row-sharded over the ranks -- data-parallel training with SyncBN, the "
                        "cluster sweep through the native sharded state machine (BASELINE.json north_star's partition); weak = "
                        "every rank holds its own dataset of --contigs contigs, shard-local sweeps")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=24_000, help="contigs of the larger CPU-baseline sample (a second one of a third of it runs beside it)")
    p.add_argument("--no-c3", action="store_true", help="skip the C3-shape leg")
    p.add_argument("--no-taxvamb", action="store_true", help="skip the joint-TaxVamb-training leg")
    p.add_argument("--no-cluster", action="store_true",
                   help="PROFILING ONLY: leave the cluster sweep out of a step (a rocprofv3 trace of a short training "
                        "run; with few epochs the latents are unstructured and the sweep degenerates).  The line is "
                        "marked as such and is not a valid headline")
    p.add_argument("--c3-epochs", type=int, default=300, help="training epochs of the C3-shape whole-job leg")
    p.add_argument("--leg-steps", type=int, default=3, help="timed whole jobs of the c1 / c3_shape legs (after one full warm-up job each)")
    p.add_argument("--deadline", type=float, default=1650.0,
                   help="seconds from process start the whole run should fit in (the driver allows 1800): only the "
                        "UNTIMED parts adapt to it (later warm-up steps run fewer epochs, the C3 leg may be skipped)")
    p.add_argument("--force-dist", action="store_true",
                   help="take the multi-GPU code path (process group + RCCL communicator) even with one rank")
    a = p.parse_args()
    world = int(os.environ.get("WORLD_SIZE", a.gpus))
    if a.config is None:
        a.config = "C3" if world > 1 else "C2"
    if a.scaling is None:
        a.scaling = "strong" if world > 1 else "weak"
    c = CONFIGS[a.config]
    a.contigs = c[0] if a.contigs is None else a.contigs
    a.samples = c[1] if a.samples is None else a.samples
    a.batch = c[2] if a.batch is None else a.batch
    a.latent = c[3] if a.latent is None else a.latent
    a.dtype = c[4] if a.dtype is None else a.dtype
    return a


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local, world


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) and relay rank 0's
    result line.  Rendezvous over 127.0.0.1 (the container hostname may not resolve)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL      # rank 0 owns stdout (the result line)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def run_step(ve, vc, lib, _lib, dl, lens, args, seed, comm=None, probe_layer=0, time_scans=False, sharded=None,
             epochs=None):
    """One pass of the hot path.  Returns per-stage seconds and counters."""
    epochs = args.epochs if epochs is None else epochs
    t0 = time.perf_counter()
    vae = ve.VAE(args.samples, nlatent=args.latent, seed=seed)
    if comm is not None:
        vae.attach_communicator(comm)
    _lib.check(lib.vh_vae_set_probe(vae._h, 1, probe_layer))
    vae._ensure_dataset(dl)
    t1 = time.perf_counter()
    vae.trainmodel(dl, nepochs=epochs, batchsteps=None)
    t2 = time.perf_counter()
    latent = vae.encode(dl)
    t3 = time.perf_counter()
    if args.no_cluster:
        ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.vh_vae_probe_result(vae._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
        return dict(setup_s=t1 - t0, train_s=t2 - t1, encode_s=t3 - t2, cluster_s=0.0, total_s=t3 - t0, clusters=0,
                    probe_ms=ms.value, probe_launches=nl.value, probe_flops=fl.value, scan_passes=0, scan_medoids=0,
                    scan_kernel_ms=0.0, scan_bytes=0, scan_resident_bytes=0, loss=vae.last_epoch_losses["loss"], latent=latent)
    if sharded is not None:
        gen = sharded(latent, lens, seed)
        backend = gen._backend
    else:
        gen = vc.ClusterGenerator(latent, lens, destroy=True, rng_seed=seed)
        backend = gen._backend
    # HIP-event timing of every scan / select kernel costs a stream synchronisation per pass: it is switched on
    # in the warm-up steps only (cluster_scan statistics), the timed steps run the sweep as a user would
    backend.set_timing(time_scans)
    n_clusters = 0
    n_points = 0
    for c in gen:
        n_clusters += 1
        n_points += len(c.members)
    t4 = time.perf_counter()
    if sharded is None:
        assert n_points == len(lens)
    if getattr(gen, "_gen", None) is not None:     # native state machine (one GPU, or sharded over the ranks): its counters
        gen._sync_native_counters()
    ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(lib.vh_vae_probe_result(vae._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
    b = backend
    L4 = (args.latent + 3) // 4 * 4
    # algorithmic bytes of a pass = N_LIVE x (4 L4 + 4 + 1): live rows at the time of the pass, not the
    # resident (dead-but-uncompacted) rows the kernels actually stream -- those are reported beside it
    live_rows = getattr(b, "live_rows_streamed", b.rows_streamed)
    out = dict(setup_s=t1 - t0, train_s=t2 - t1, encode_s=t3 - t2, cluster_s=t4 - t3, total_s=t4 - t0,
               clusters=n_clusters, probe_ms=ms.value, probe_launches=nl.value, probe_flops=fl.value,
               scan_passes=b.scan_passes, scan_medoids=b.scan_medoids, scan_kernel_ms=b.kernel_ms,
               scan_bytes=live_rows * (4 * L4 + 5), scan_resident_bytes=b.rows_streamed * (4 * L4 + 5),
               loss=vae.last_epoch_losses["loss"], latent=latent)
    b.close()
    return out


def probe_roofline(steps, what, peak):
    ms = sum(r["probe_ms"] for r in steps)
    n = sum(r["probe_launches"] for r in steps)
    if not n:
        return None
    flops = steps[-1]["probe_flops"]
    ach = flops / (ms / n * 1e-3) / 1e12
    return {"kernel": what, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "avg_launch_ms": ms / n, "launches": n, "flops_per_launch": flops,
            "timing": "kernel begin/end timestamps (hipExtLaunchKernelGGL start/stop events on the launching stream)"}


def scan_summary(timed, wall, measured_in):
    """cluster_scan object: `timed` = steps that ran with HIP-event timing of every scan / select kernel (kernel-time
    fraction), `wall` = steps whose sweep ran untimed, as a user runs it (fraction over the sweep's wall time)."""
    kms = sum(r["scan_kernel_ms"] for r in timed)
    kbytes = sum(r["scan_bytes"] for r in timed)
    wbytes = sum(r["scan_bytes"] for r in wall)
    wres = sum(r["scan_resident_bytes"] for r in wall)
    wsec = sum(r["cluster_s"] for r in wall)
    return {
        "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBPS,
        "bytes": "algorithmic: live rows at the time of each pass x (4 L4 + 4 + 1) B (SURVEY.md 8d)",
        "achieved": kbytes / (kms * 1e-3) / 1e9 if kms else None,
        "frac": kbytes / (kms * 1e-3) / 1e9 / PEAK_HBM_GBPS if kms else None,
        "achieved_over_sweep_wall_time": wbytes / wsec / 1e9 if wsec else None,
        "frac_over_sweep_wall_time": wbytes / wsec / 1e9 / PEAK_HBM_GBPS if wsec else None,
        "resident_over_live_bytes": wres / wbytes if wbytes else None,
        "passes": sum(r["scan_passes"] for r in (wall or timed)), "medoids": sum(r["scan_medoids"] for r in (wall or timed)),
        "kernel_ms_total": kms, "sweep_wall_s": wsec,
        "measured_in": measured_in,
    }


def pmc_traffic(tag, want_source=False):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (profiles/), or None."""
    for rnd in ("r06f6", "r06f5", "r06f4", "r06zzz", "r06zz", "r06y", "r05z", "r04t", "r03"):   # the latest committed passes first
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_roofline_{tag}.json")
        try:
            with open(path) as fh:
                v = json.load(fh).get("hbm_bytes_per_launch")
                return (v, os.path.relpath(path, ROOT)) if want_source else v
        except (OSError, ValueError):
            continue
    return (None, None) if want_source else None


def config_leg(args, name, ve, vc, lib, _lib, synth, epochs, steps=3, warmup=1, est_job_s=None):
    """The WHOLE job at another BASELINE configuration on this GPU with the headline's protocol: `warmup` untimed full jobs, then
    EXACTLY `steps` timed jobs (VAE.trainmodel with `epochs`, default the CLI's 300 -> encode -> full cluster sweep) bracketed by
    a device synchronisation on both sides; then the sweep once more on the last latents with HIP-event timing of every pass
    (its kernel-time roofline).  "C3": 2 M contigs x 1000 samples (8.8 GB of features), the shape BASELINE's 10x target is
    quoted on, in the headline's dtype.  "C1": configs[1], 200 k x 50, batch 4096, fp32 MFMA.  When the run's --deadline cannot
    hold warmup + steps jobs, fewer are run (never fewer than one timed step) and the object says so."""
    n, S, bs, lat_w, cfg_dtype = CONFIGS[name]
    dtype = cfg_dtype if name == "C1" else args.dtype
    a3 = argparse.Namespace(**vars(args))
    a3.contigs, a3.samples, a3.batch, a3.latent, a3.epochs, a3.no_cluster, a3.dtype = n, S, bs, lat_w, epochs, False, dtype
    ve.set_compute_dtype(dtype)
    planned = (warmup, steps)
    try:
        t0 = time.perf_counter()
        ab, tnf, lens, _ = synth.features(n, S, seed=3)
        t_synth = time.perf_counter() - t0
        t0 = time.perf_counter()
        dl = ve.make_dataloader(ab, tnf, lens, batchsize=bs, destroy=True)
        t_prep = time.perf_counter() - t0
        prep_on_device = getattr(dl.dataset, "_vambhip_prepared", None) is not None
        del ab, tnf
        # allocations / kernel attributes of this shape, outside the timed steps
        run_step(ve, vc, lib, _lib, dl, lens, argparse.Namespace(**dict(vars(a3), no_cluster=True)), seed=1003, epochs=1)
        if est_job_s is not None:   # fit the plan into what is left of --deadline (untimed parts first)
            left = args.deadline - (time.perf_counter() - T_START) - 60.0
            while warmup > 0 and (warmup + steps) * est_job_s > left:
                warmup -= 1
            while steps > 1 and (warmup + steps) * est_job_s > left:
                steps -= 1
        warm_jobs = []
        for i in range(warmup):
            tw = time.perf_counter()
            run_step(ve, vc, lib, _lib, dl, lens, a3, seed=2003 + i, probe_layer=0, time_scans=False).pop("latent")
            warm_jobs.append(time.perf_counter() - tw)
        _lib.check(lib.vh_device_synchronize())
        t0 = time.perf_counter()
        runs = []
        for i in range(steps):
            ts = time.perf_counter()
            r = run_step(ve, vc, lib, _lib, dl, lens, a3, seed=3 + i, probe_layer=0, time_scans=False)
            r["job_s"] = time.perf_counter() - ts
            if i + 1 < steps:
                r.pop("latent")
            runs.append(r)
        _lib.check(lib.vh_device_synchronize())
        t_all = time.perf_counter() - t0
    finally:
        ve.set_compute_dtype(args.dtype)
    r = runs[-1]
    latent = r.pop("latent")
    # the same sweep with per-pass kernel timing (a stream synchronisation per pass: not part of the timed jobs)
    # (the timed job's generator normalised `latent` in place: destroy=True -- normalising twice is not bit-idempotent)
    gen = vc.ClusterGenerator(latent.copy(), lens, destroy=True, normalized=True, rng_seed=3 + steps - 1)
    gen._backend.set_timing(True)
    tt = time.perf_counter()
    n_clusters2 = sum(1 for _ in gen)
    t_sweep2 = time.perf_counter() - tt
    gen._sync_native_counters()
    bk = gen._backend
    L4 = (lat_w + 3) // 4 * 4
    timed = dict(scan_kernel_ms=bk.kernel_ms, scan_bytes=getattr(bk, "live_rows_streamed", bk.rows_streamed) * (4 * L4 + 5),
                 scan_resident_bytes=bk.rows_streamed * (4 * L4 + 5), scan_passes=bk.scan_passes, scan_medoids=bk.scan_medoids,
                 cluster_s=t_sweep2)
    bk.close()
    D = S + NTNF + 1
    flops_contig = 12 * HIDDEN * (D + HIDDEN + lat_w) - 2 * D * HIDDEN
    bf16 = dtype == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    mean = lambda k: float(np.mean([q[k] for q in runs]))   # noqa: E731
    t_epoch = mean("train_s") / a3.epochs
    roof = probe_roofline(runs, f"first encoder layer, M={bs}, K=D={D} (padded {(D + 31) // 32 * 32}), N=512, "
                                f"{'bf16 MFMA' if bf16 else 'fp32 MFMA'}", peak)
    return {"workload": f"{name} on one GPU: {n} contigs x {S} samples (D={D}), batch {bs}, {'bf16' if bf16 else 'f32'}; whole job = "
                        f"{a3.epochs} train epochs + encode + full cluster sweep; {len(warm_jobs)} untimed warm-up job(s), then "
                        f"{steps} timed jobs between two device synchronisations",
            "dtype": "bf16" if bf16 else "f32",
            "value": n * steps / t_all, "unit": "contigs/s", "steps": steps, "warmup": len(warm_jobs),
            "planned_warmup_steps": list(planned), "ms_per_step": t_all / steps * 1e3,
            "job_s": t_all / steps, "job_s_each": [q["job_s"] for q in runs], "warmup_job_s": warm_jobs,
            "train_s": mean("train_s"), "encode_s": mean("encode_s"), "cluster_s": mean("cluster_s"), "setup_s": mean("setup_s"),
            "clusters": [q["clusters"] for q in runs], "clusters_second_sweep": n_clusters2, "final_loss": r["loss"],
            "epoch_ms": t_epoch * 1e3, "us_per_step": t_epoch / (n // bs) * 1e6,
            "train_contigs_per_s_per_epoch": n / t_epoch, "train_tflops_algorithmic": flops_contig * n / t_epoch / 1e12,
            "synthetic_input_s": t_synth, "make_dataloader_s": t_prep,
            "make_dataloader_on": "device (csrc/prep.hip: one upload of the raw matrices + normalisation kernels, "
                                  "PCIe-inclusive)" if prep_on_device else "host (numpy)",
            "roofline_encoder_gemm": roof,
            "cluster_scan": scan_summary([timed], runs, "kernel time: one more, event-timed sweep over the last job's latents; "
                                                          "wall time: the sweeps of the timed jobs")}


def taxvamb_leg(ve, synth, n=200_000, S=50, n_nodes=1000, batch=256, epochs=3):
    """Row N4's trainer in front of the driver: TaxVamb's joint training (VAEVAEHLoss.trainmodel: three networks, seven passes and
    one Adam step per batch, hierarchical loss over a taxonomy; vamb/__main__.py:1988-2047) at the CLI's starting batch size on
    a synthetic problem, then VAEJoint.encode.  fp32 step.  Not part of `value`."""
    from vamb_amd import taxvamb_encode as vt

    rng = np.random.RandomState(0)
    parents = [-1] + [int(rng.randint(max(0, i - 60), i)) for i in range(1, n_nodes)]
    ab, tnf, lens, genome = synth.features(n, S, seed=3)
    nodes = (genome.astype(np.int64) * 7919) % n_nodes
    names = [f"n{i}" for i in range(n_nodes)]
    dl_v = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=batch)
    dl_j = vt.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, nodes, n_nodes, parents, batchsize=batch)
    dl_l = vt.make_dataloader_labels_hloss(ab, tnf, lens, nodes, n_nodes, parents, batchsize=batch)
    dl = vt.make_dataloader_semisupervised_hloss(dl_j, dl_v, dl_l, n_nodes, parents, (S, NTNF, 1, n_nodes), 0, batchsize=batch)
    vae = vt.VAEVAEHLoss(S, n_nodes, names, parents)
    vae.trainmodel(dl, nepochs=1, batchsteps=None)      # uploads, first launches
    first = vae.last_epoch_metrics["loss"]
    t0 = time.perf_counter()
    vae.trainmodel(dl, nepochs=epochs, batchsteps=None)
    t_epoch = (time.perf_counter() - t0) / epochs
    t0 = time.perf_counter()
    latent = vae.VAEJoint.encode(dl_j)
    t_enc = time.perf_counter() - t0
    steps = n // batch
    return {"workload": f"joint TaxVamb training (VAEVAEHLoss: VAEVamb + VAELabels + VAEJoint, hidden 512-512, latent 32, flat-softmax "
                        f"loss over a {n_nodes}-node taxonomy): {n} contigs x {S} samples, batch {batch} (the CLI's starting batch), "
                        f"{epochs} timed epochs after one warm-up epoch; f32",
            "dtype": "f32", "epoch_s": t_epoch, "ms_per_step": 1e3 * t_epoch / steps, "steps_per_epoch": steps,
            "passes_per_step": 7, "train_contigs_per_s_per_epoch": steps * batch / t_epoch, "encode_s": t_enc,
            "loss_first_epoch": first, "loss_last_epoch": vae.last_epoch_metrics["loss"],
            "latent_finite": bool(np.isfinite(latent).all())}


def _cpu_sweep_only(n, latent, lens, co):
    """The cluster restatement alone on the first n GPU latents (the super-linear term of the job: one more point for its exponent)."""
    t0 = time.perf_counter()
    nclu = sum(1 for _ in co.OracleClusterGenerator(np.ascontiguousarray(latent[:n]), lens[:n], rng_seed=1))
    return dict(contigs=n, threads=1, epoch_s=None, encode_s=None, cluster_s=time.perf_counter() - t0, clusters=nclu, job_s=None,
                sweep_only=True)


def _cpu_sample(args, n, latent, lens, threads, co, vo, ve, synth, cpu_epochs=3, with_cluster=True):
    """The oracle port on the first n contigs of the workload: seconds per training epoch, for the encode pass and (with_cluster)
    for the full cluster sweep of the first n GPU latents.  The caller has limited the BLAS pool to `threads`."""
    ab, tnf, ln, _ = synth.features(n, args.samples, seed=101)
    dl = ve.make_dataloader(ab, tnf, ln, batchsize=args.batch, destroy=True, _prep="host")   # the CPU baseline never touches the GPU
    d, t, a, w = (x.numpy() for x in dl.dataset.tensors)
    st = vo.init_state(args.samples, [HIDDEN, HIDDEN], args.latent, 1)
    m = vo.OracleVAE(args.samples, [HIDDEN, HIDDEN], args.latent, None, 200.0, 0.2, state=st, dtype=np.float32)
    rng = np.random.RandomState(0)
    bs = min(args.batch, n)
    nb = max(1, n // bs)
    wrows = np.arange(bs)    # one untimed step: BLAS thread pool, page faults of the work arrays
    m.train_step(d[wrows], t[wrows], a[wrows], w[wrows], rng.standard_normal((bs, args.latent)).astype(np.float32),
                 [rng.random_sample((bs, HIDDEN)) >= 0.2 for _ in range(4)])
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
    t_clu, nclu = None, None
    if with_cluster:
        t0 = time.perf_counter()
        nclu = sum(1 for _ in co.OracleClusterGenerator(np.ascontiguousarray(latent[:n]), lens[:n], rng_seed=1))
        t_clu = time.perf_counter() - t0
    return dict(contigs=n, batch=bs, threads=int(threads), epoch_s=t_epoch, encode_s=t_enc, cluster_s=t_clu, clusters=nclu,
                job_s=None if t_clu is None else args.epochs * t_epoch + t_enc + t_clu, epochs_timed=cpu_epochs)


def cpu_baseline(args, latent, lens):
    """The oracle (numpy VAE restatement + C cluster restatement) on THREE bounded samples of the workload (n, n / 3, n / 9
    contigs) with the reference's default thread count (min(cores, 8), vamb/__main__.py:27-28), and the largest sample's training
    and encode once more with ALL visible cores (SURVEY.md 8d asks for both; the cluster restatement is scalar code, one
    thread, as the reference's sweep is one torch thread pool over a [n] vector).  Training and encoding cost is linear in the
    number of contigs; the cluster sweep is not (every cluster costs passes over all remaining contigs), so its exponent is
    fitted through the three samples and the extrapolation to the full workload is stated explicitly.  `value` is the measured
    throughput of the LARGEST sample at the reference's default thread count (no extrapolation)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cluster_oracle as co
    import vae_oracle as vo
    from vamb_amd import encode as ve, synth

    cores = os.cpu_count() or 1
    try:
        cores_avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores_avail = cores
    threads = min(8, cores_avail)
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None

    def limited(nthreads, fn):
        if threadpool_limits is None:
            return fn()
        with threadpool_limits(limits=nthreads):
            return fn()

    n_big = min(args.cpu_sample, args.contigs, len(latent) // 2 if len(latent) >= 2 * args.cpu_sample else len(latent))
    sizes = sorted({max(1000, n_big // 9), max(1000, n_big // 3), n_big})
    samples = [limited(threads, lambda n=n: _cpu_sample(args, n, latent, lens, threads, co, vo, ve, synth)) for n in sizes]
    big = samples[-1]
    # a fourth, larger point for the sweep's exponent: 2 n contigs, cluster restatement only
    sweep_points = list(samples)
    if len(latent) >= 2 * n_big and 2 * n_big <= args.contigs:
        sweep_points.append(_cpu_sweep_only(2 * n_big, latent, lens, co))
    # the largest sample's BLAS-bound stages with every visible core (the sweep term is the one measured above)
    all_cores = None
    if cores_avail > threads:
        ac = limited(cores_avail, lambda: _cpu_sample(args, n_big, latent, lens, cores_avail, co, vo, ve, synth, cpu_epochs=2,
                                                      with_cluster=False))
        ac["cluster_s"], ac["clusters"] = big["cluster_s"], big["clusters"]
        ac["job_s"] = args.epochs * ac["epoch_s"] + ac["encode_s"] + ac["cluster_s"]
        ac["value"] = n_big / ac["job_s"]
        ac["note"] = (f"training / encode of the {n_big}-contig sample with a BLAS pool of {cores_avail} threads (numpy fp32 GEMMs of "
                      f"{big['batch']} x 512: more threads than the reference's default do not pay at this size); the sweep term is the "
                      "single-threaded one of the 8-thread run")
        all_cores = ac
    # cluster sweep: t = c n^p, least squares through the samples; training / encoding: linear through the largest sample
    if len(sweep_points) > 1:
        p_clu = float(np.polyfit(np.log([q["contigs"] for q in sweep_points]), np.log([q["cluster_s"] for q in sweep_points]), 1)[0])
    else:
        p_clu = 1.0
    scale = args.contigs / n_big
    top = sweep_points[-1]   # the sweep is extrapolated from the LARGEST point measured
    full = dict(contigs=args.contigs, train_s=args.epochs * big["epoch_s"] * scale, encode_s=big["encode_s"] * scale,
                cluster_s=top["cluster_s"] * (args.contigs / top["contigs"]) ** p_clu)
    full["job_s"] = full["train_s"] + full["encode_s"] + full["cluster_s"]
    full["contigs_per_s"] = args.contigs / full["job_s"]
    calib = None
    try:
        with open(os.path.join(ROOT, "oracle", "cpu_calibration.json")) as fh:
            c = json.load(fh)
        calib = {"reference_over_port": c["reference_over_port"], "measured_on": f"{c['cpu']}, {c['threads']} threads",
                 "sample": c["sample"],
                 "points": [{"contigs": q.get("contigs"), "reference_over_port": q["reference_over_port"]} for q in c.get("points", [])],
                 "cluster_time_exponent_on_that_machine": c.get("cluster_time_exponent"),
                 "note": "the real reference (vamb/{encode,cluster}.py, torch CPU) and this port timed on the same samples "
                         "in the build container (oracle/calibrate_cpu_baseline.py: a DIFFERENT machine from the one this "
                         "line was measured on); > 1 means the reference is slower"}
    except (OSError, ValueError, KeyError):
        pass
    ref_est = None if calib is None else n_big / (big["job_s"] * calib["reference_over_port"]["job_300_epochs"])
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
    except OSError:
        pass
    # `value` is the LIKE-FOR-LIKE figure: the port's throughput on the benchmarked workload (training / encode scaled linearly from
    # the largest sample, the sweep by the fitted power law) -- the number a GPU-over-CPU ratio may be formed with; what was
    # MEASURED without extrapolation is `value_on_sample` (until round 6 `value` was the sample's figure).
    return dict(value=full["contigs_per_s"], unit="contigs/s (extrapolated to the benchmarked workload; value_on_sample = measured)",
                value_on_sample=n_big / big["job_s"], cores=int(threads), kind="port",
                sample=(f"{n_big} contigs x {args.samples} samples, batch {big['batch']}: {big['epochs_timed']} oracle epochs timed "
                        f"({big['epoch_s']:.3f} s/epoch, numpy fp32 BLAS, {threads} threads) extrapolated to {args.epochs}, + encode "
                        f"{big['encode_s']:.3f} s + full cluster sweep of the first {n_big} GPU latents {big['cluster_s']:.3f} s "
                        f"({big['clusters']} clusters, scalar C, 1 thread); samples of {', '.join(str(q['contigs']) for q in samples[:-1])} "
                        f"contigs beside it; the BLAS-bound stages again with all {cores_avail} visible cores (all_cores)"),
                epoch_s=big["epoch_s"], encode_s=big["encode_s"], cluster_s=big["cluster_s"],
                samples=sweep_points, cluster_time_exponent=p_clu, cluster_time_exponent_points=len(sweep_points), all_cores=all_cores,
                host={"cpu": cpu_model, "cores_visible": int(cores_avail), "cores_online": int(cores), "threads_used": int(threads)},
                extrapolated_to_workload={**full, "how": f"training and encode linear in contigs from the {n_big}-contig sample; cluster "
                                          f"sweep t = c n^p with p = {p_clu:.2f} fitted through the {len(sweep_points)} samples, from the largest of them (a sweep is "
                                          "super-linear: every cluster costs passes over all remaining contigs); an estimate, not a "
                                          "measurement"},
                calibration_vs_reference=calib, reference_estimate_contigs_per_s=ref_est)


def _cluster_order(_lib):
    """Which arithmetic the timed sweeps ran in (library option scan.reference_order)."""
    _lib.sync_env_options()
    mode = _lib.get_option("scan.reference_order", 2)
    what = {2: "reference order (default): matmul / norm evaluated as the reference's own torch / oneMKL AVX-512 CPU build does; the "
               "tuned scan kernels filter with the ascending fmaf chain and re-evaluate pairs at decision boundaries in that order. "
               "Bit-exact cluster streams against the real reference on every golden fixture (tests/test_cluster_gpu.py)",
            1: "reference order on the plain one-pair-per-lane kernel (cross-check mode)",
            0: "ascending fmaf chain (default of rounds 1-3; NOT the bit-exact mode)"}[int(mode)]
    return {"scan.reference_order": int(mode), "meaning": what}


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
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    claim_stdout()
    rank, local, world = dist_env()
    args.gpus = world
    from vamb_amd import _lib, cluster as vc, encode as ve, synth

    ve.set_compute_dtype(args.dtype)
    os.environ.pop("VAMBHIP_PRECISION", None)     # --dtype decides
    lib = _lib.load()
    _lib.require_gpu()
    _lib.check(lib.vh_set_device(local % max(1, _lib.device_count())))   # (several ranks may share a GPU in a functional run)
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

        # VAMBHIP_BENCH_HOST_PLANE=1: the library's collectives through this gloo group instead of RCCL -- a FUNCTIONAL run of the
        # N > 1 path with several ranks on one GPU (RCCL refuses two ranks per device); never a performance number
        host_plane = bool(os.environ.get("VAMBHIP_BENCH_HOST_PLANE"))
        comm = parallel.Communicator.from_torch_distributed(dist, rccl="host" if host_plane else True)

    strong = args.scaling == "strong" and (world > 1 or args.force_dist)   # (--force-dist: the multi-GPU code path on one rank)
    if strong:
        # ONE dataset, row-sharded: rank r holds rows [r n / world, (r + 1) n / world)
        ab, tnf, lens_all, _ = synth.features(args.contigs, args.samples, seed=1)
        lo, hi = rank * args.contigs // world, (rank + 1) * args.contigs // world
        ve.set_prep_mode("host")     # the normalised tensors are sliced on the host below
        # strong scaling = the SAME job on more GPUs: the optimisation batch stays --batch rows in total (every rank contributes
        # batch / N rows per step, north_star: "training mini-batches ... partition across the 8 GPUs"), so an epoch has the
        # same number of optimiser steps as on one GPU
        if args.batch % world != 0:
            raise SystemExit(f"--batch {args.batch} is not divisible by {world} GPUs")
        dl_full = ve.make_dataloader(ab, tnf, lens_all, batchsize=args.batch, destroy=True)
        ve.set_prep_mode("auto")
        import torch

        tens = [t[lo:hi].contiguous() for t in dl_full.dataset.tensors]
        dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(*tens), batch_size=args.batch,
                                         shuffle=True, drop_last=True)
        del dl_full, ab, tnf
        lens = lens_all[lo:hi]
        from vamb_amd import parallel as _par

        def sharded(latent, lens_local, seed):
            return _par.sharded_cluster_generator(comm, latent, lens_local, destroy=True, rng_seed=seed)
    else:
        # synthetic inputs of the named shape (per-rank shard under weak scaling), normalised on the host
        # exactly as `vamb bin default` does, then uploaded once: resident in HBM before the clock starts
        ab, tnf, lens, _ = synth.features(args.contigs, args.samples, seed=1 + rank)
        # under data parallelism the loader's batch size is the ALL-RANK batch: --batch rows per GPU
        t_prep0 = time.perf_counter()
        dl = ve.make_dataloader(ab, tnf, lens, batchsize=args.batch * world, destroy=True)
        prep_s = time.perf_counter() - t_prep0
        prep_on_device = getattr(dl.dataset, "_vambhip_prepared", None) is not None
        del ab, tnf
        sharded = None

    def barrier():
        _lib.check(lib.vh_device_synchronize())
        if dist is not None:
            dist.barrier()
            _lib.check(lib.vh_device_synchronize())

    warm = []
    warm_epochs = []
    for i in range(args.warmup):
        # the warm-up steps time a hidden 512x512 layer instead (roofline_hidden) and, in the first one, the scan
        # kernels.  The first warm-up step is a full step; if (warmup + steps) full steps would not fit --deadline,
        # the remaining warm-up steps (untimed by definition) run a tenth of the epochs -- reported as warmup_epochs.
        ep = args.epochs
        if i > 0:
            t_full = warm[0]["total_s"]
            projected = (time.perf_counter() - T_START) + (args.warmup - i + args.steps) * t_full + 200.0
            if projected > args.deadline:
                ep = max(1, args.epochs // 10)
        if dist is not None:     # every rank must take the same decision
            import torch

            e = torch.tensor([ep], dtype=torch.int64)
            dist.all_reduce(e, op=dist.ReduceOp.MIN)
            ep = int(e.item())
        warm_epochs.append(ep)
        # a shortened warm-up step leaves the cluster sweep out: after a tenth of the epochs the latents are barely
        # structured and the sweep degenerates into 10^5..10^6 tiny clusters (minutes instead of seconds)
        short = ep < args.epochs
        if short:
            args.no_cluster, keep = True, args.no_cluster
        warm.append(run_step(ve, vc, lib, _lib, dl, lens, args, seed=1000 + i, comm=comm, probe_layer=1,
                             time_scans=(i == 0), sharded=sharded, epochs=ep))
        if short:
            args.no_cluster = keep
    barrier()
    t0 = time.perf_counter()
    results = []
    for i in range(args.steps):
        results.append(run_step(ve, vc, lib, _lib, dl, lens, args, seed=i, comm=comm, probe_layer=0,
                                time_scans=args.warmup == 0, sharded=sharded))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        K = max(1, args.steps)
        job_contigs = args.contigs if strong else args.contigs * world
        D = args.samples + NTNF + 1
        bf16 = args.dtype == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        shape = (args.contigs, args.samples, args.batch, args.latent, args.dtype)
        cfg_name = next((k for k, v in CONFIGS.items() if v == shape), "custom")
        scan_src = warm[:1] if warm else results      # the steps that ran with scan-kernel timing on
        scan_ms = sum(r["scan_kernel_ms"] for r in scan_src)
        scan_bytes = sum(r["scan_bytes"] for r in scan_src)
        arith = "bf16 storage + bf16 MFMA / fp32 accumulate" if bf16 else "fp32 MFMA"
        gemm_kind = "gemm_bf16_kernel<128,128,2x4 waves,E16_HIDDEN_TRAIN>" if bf16 else "gemm_f32_kernel<64,64,2x2 waves,EPI_HIDDEN_TRAIN>"
        roof = probe_roofline(results, f"{gemm_kind}: encoder layer 0, M=batch={args.batch}, K=D={D}, N=512 "
                                       "(bias + leaky-relu + dropout + BatchNorm batch sums in the epilogue)", peak)
        if roof is not None:
            roof["traffic"], src = pmc_traffic(cfg_name.lower(), want_source=True)
            roof["traffic_source"] = (f"static: {src} (rocprofv3 --pmc passes of this kernel at this shape, profiles/run_r06_final6.sh "
                                      "for r06f6 / run_r06_final5.sh for r06f5: FETCH_SIZE doubled as the gfx950 guide prescribes + WRITE_SIZE, "
                                      "separate passes); NOT counted in this run")
            if roof["traffic"]:
                # the same launches against the other roofline: at C2 the kernel's arithmetic intensity (flops / PMC bytes)
                # is below the ridge of 2.5 PFLOP/s : 8 TB/s = 312 flop/B, i.e. it is the HBM side that binds there
                gbps = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
                roof["hbm_side"] = {"achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                    "flop_per_byte": roof["flops_per_launch"] / roof["traffic"], "ridge_flop_per_byte":
                                    peak * 1e12 / (PEAK_HBM_GBPS * 1e9)}
        line = {
            "metric": "contigs/sec through VAE-train+encode+cluster; VAE epoch step time",
            "value": job_contigs * K / elapsed,
            "unit": "contigs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": ("strong" if strong else "weak") if world > 1 else None,   # (one GPU: nothing is scaled)
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{cfg_name}: {args.contigs} contigs x {args.samples} samples "
                             f"{'in total' if strong else 'per GPU'} (D={D}), hidden 512-512, latent {args.latent}, "
                             f"batch {str(args.batch) + ' in total (' + str(args.batch // world) + ' per GPU)' if strong else str(args.batch) + ' per GPU'}, {arith}; {args.epochs} train epochs"
                             f"{' (reference CLI default)' if args.epochs == 300 else ' (reference CLI default is 300)'}"
                             f" + encode + {'NO cluster sweep (--no-cluster: profiling run, not a headline)' if args.no_cluster else 'full cluster sweep'} per step; features resident in HBM before the clock starts "
                             "(make_dataloader -- one H2D upload + normalisation -- outside the timed region)"),
                "contigs_per_gpu": args.contigs if not strong else args.contigs // world, "samples": args.samples,
                "batch": args.batch, "batch_per_gpu": args.batch // world if strong else args.batch, "epochs": args.epochs,
                "parallelism": (f"dp{world}" + ("+sharded-cluster" if strong else "")) if world > 1 else "single",
            },
            "roofline": roof,
            "roofline_hidden": probe_roofline(warm, f"{gemm_kind}: encoder layer 1, M=batch={args.batch}, K=512, N=512 "
                                                    "(the GEMM shape that dominates the step), warm-up steps", peak),
            "epoch_ms": float(np.mean([r["train_s"] for r in results]) / args.epochs * 1e3),
            "us_per_train_step": float(np.mean([r["train_s"] for r in results]) / args.epochs /
                                       max(1, args.contigs // args.batch) * 1e6),
            "train_contigs_per_s_per_epoch": float(job_contigs / (np.mean([r["train_s"] for r in results]) / args.epochs)),
            "train_s": float(np.mean([r["train_s"] for r in results])),
            "encode_ms": float(np.mean([r["encode_s"] for r in results]) * 1e3),
            "cluster_ms": float(np.mean([r["cluster_s"] for r in results]) * 1e3),
            "clusters_per_step": int(np.mean([r["clusters"] for r in results])),
            "cluster_scan": scan_summary(scan_src, results, ("kernel time: first warm-up step (HIP-event timing of every pass); " if warm else "kernel time: timed steps; ") + "wall time: the timed steps"),
            "cluster_order": _cluster_order(_lib),
            "multi_gpu": None if comm is None else {
                **comm.info(),
                "rccl_ranks": comm.info()["reported_ranks"] if comm.info()["data_plane"] == "rccl" else None,
                "training": "data parallel: every rank holds a row shard and contributes batch / N rows per step; loss normalised by "
                            "the all-rank batch, BatchNorm statistics synchronised (fp64 sums all-reduced), flat gradient all-reduced "
                            "over RCCL in two buckets (decoder side under the encoder's backward)",
                "cluster": ("ONE latent matrix row-sharded over the ranks, the native sharded state machine (vh_gen_create_sharded): "
                            "every rank scans its shard, one all-gather of exact integer accumulators + list parts per pass"
                            if strong else "shard-local sweeps (weak scaling: N independent datasets)"),
                "n1_reference": ("the c3_shape object of the N = 1 line is this workload on one GPU: compare this line's `value` with that "
                                 "object's `value`, not with the N = 1 line's `value` (C2, another workload)" if cfg_name == "C3" and strong
                                 else None)},
            "make_dataloader": None if strong else {
                "seconds": prep_s, "on": "device" if prep_on_device else "host",
                "note": "outside the timed region; on the device it is ONE upload of the raw abundance / TNF matrices "
                        "(PCIe-inclusive) + the normalisation kernels of csrc/prep.hip, bit-identical to the host numpy path"},
            "final_loss": results[-1]["loss"] if results else None,
            "warmup_epochs": warm_epochs,
        }
        latent_keep = results[-1]["latent"][: 2 * args.cpu_sample].copy() if results else None   # (2 x: the sweep-only fourth point)
        lens_keep = lens[: 2 * args.cpu_sample].copy()
        for r in results + warm:
            r.pop("latent", None)
        if not args.no_cpu_baseline and world == 1 and latent_keep is not None and not args.no_cluster:
            line["cpu_baseline"] = cpu_baseline(args, latent_keep, lens_keep)
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_taxvamb:      # ~10 s
            if args.deadline - (time.perf_counter() - T_START) < 60.0:
                line["taxvamb"] = {"skipped": "not enough of --deadline left for the extra leg"}
            else:
                try:
                    ve.set_compute_dtype("fp32")
                    line["taxvamb"] = taxvamb_leg(ve, synth)
                except Exception as e:   # never lose the headline because of an extra leg
                    line["taxvamb"] = {"error": repr(e)}
                finally:
                    ve.set_compute_dtype(args.dtype)
        if world == 1 and not args.no_c3 and cfg_name != "C3":
            del dl      # free the headline's dataset first (the device copy cached on the loader)
            if cfg_name != "C1":     # configs[1] (200 k x 50, batch 4096, fp32): ~3 s of input + ~6 s job + ~1 s timed sweep
                if args.deadline - (time.perf_counter() - T_START) < 60.0:
                    line["c1"] = {"skipped": "not enough of --deadline left for the extra leg"}
                else:
                    try:
                        line["c1"] = config_leg(args, "C1", ve, vc, lib, _lib, synth, args.c3_epochs, steps=args.leg_steps, warmup=1,
                                                est_job_s=8.0)
                    except Exception as e:   # never lose the headline because of an extra leg
                        line["c1"] = {"error": repr(e)}
            # synthetic input ~35 s + upload + (warm-up + timed) jobs of ~1.1 x the headline's step + event-timed sweep ~30 s + margin
            est_c3 = 1.1 * elapsed / K * (args.c3_epochs / max(1, args.epochs)) if not args.no_cluster else 40.0
            if args.deadline - (time.perf_counter() - T_START) < 150.0 + est_c3:
                line["c3_shape"] = {"skipped": "not enough of --deadline left for the extra leg"}
            else:
                try:
                    line["c3_shape"] = config_leg(args, "C3", ve, vc, lib, _lib, synth, args.c3_epochs, steps=args.leg_steps, warmup=1,
                                                  est_job_s=est_c3 + 10.0)
                except Exception as e:
                    line["c3_shape"] = {"error": repr(e)}
        emit_result(line)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
