"""Two processes on ONE GPU: the row-sharded cluster sweep with the product's HIP scan backend on every shard and the gloo
control plane (Communicator(rccl=False): RCCL refuses two ranks on the same device, so the device data plane of
vh_clu_scan_sharded stays covered by tests/test_dp_gpu.py with one rank).  What runs here is exactly what runs on N GPUs with
`rccl=False` -- ShardedScanBackend over HipScanBackend shards, the per-pass exchange of query vectors, exact int64
accumulators and selected rows -- and the stream must be the reference's golden single-process stream, bit for bit."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import test_parallel_gloo as pg  # noqa: E402  (process-spawning helpers)

pytestmark = pytest.mark.gpu


def _sharded_cluster_hip(comm):
    import fixture_defs as fd
    from vamb_amd import _lib, cluster as vc, parallel

    _lib.require_gpu()
    vc.ClusterGenerator.PACK_MIN_ROWS = 64   # make the lazy packing happen on small fixtures
    out = {}
    for name in ("blob_s008_n2000", "blob_s050_window", "blob_zero_dup", "blob_s050_n3000"):
        mat, lens, kw = fd.cluster_inputs(name)
        cut = pg.uneven_cuts(len(mat), comm.world)
        lo, hi = cut[comm.rank], cut[comm.rank + 1]
        gen = parallel.sharded_cluster_generator(comm, mat[lo:hi].copy(), lens[lo:hi], **kw)   # default factory: HipScanBackend
        assert type(gen._backend.local).__name__ == "HipScanBackend"
        got = fd.pack_stream(list(gen))
        out[name] = fd.streams_equal(got, fd.load("cluster_" + name), pvr_rtol=1e-6)
        gen._backend.local.close()
    return out


def _sharded_cluster_native(comm_control):
    """The NATIVE sharded state machine (vh_gen_create_sharded) on two uneven HIP shards of one GPU: its collectives (ONE
    all-gather of accumulators + list parts per pass, the select gathers, the one-off gather of the normalised matrix) run
    over the library's HOST data plane, i.e. over this gloo group -- RCCL refuses two ranks on one device."""
    import fixture_defs as fd
    from vamb_amd import _lib, cluster as vc, parallel

    _lib.require_gpu()
    vc.ClusterGenerator.PACK_MIN_ROWS = 64   # make the lazy packing happen on small fixtures
    comm = parallel.Communicator(comm_control.dist, rccl="host")
    world = comm_control.world
    assert comm.info() == {"rank": comm.rank, "world": world, "reported_ranks": world, "data_plane": "host"}
    out = {}
    for name in ("blob_s008_n2000", "blob_s050_window", "blob_zero_dup", "blob_s050_n3000", "blob_s008_n10000", "test_cluster_py"):
        mat, lens, kw = fd.cluster_inputs(name)
        # the one-off gather of the normalised matrix goes through a bounded staging buffer: 8 KiB here, i.e. dozens of chunks
        # with ragged last ones on these fixtures (the default, 64 MiB, would take each of them in one)
        if name in ("blob_s050_window", "blob_s008_n10000"):
            os.environ["VAMBHIP_GATHER_STAGE_BYTES"] = "8192"
        else:
            os.environ.pop("VAMBHIP_GATHER_STAGE_BYTES", None)
        cut = pg.uneven_cuts(len(mat), world)
        lo, hi = cut[comm.rank], cut[comm.rank + 1]
        gen = parallel.sharded_cluster_generator(comm, mat[lo:hi].copy(), lens[lo:hi], **kw)
        assert gen._sharded_native and gen._gen is not None
        got = fd.pack_stream(list(gen))
        out[name] = fd.streams_equal(got, fd.load("cluster_" + name), pvr_rtol=1e-6)
        gen._backend.local.close()
    comm.close()
    return out


def _dp_training_host_plane(comm_control):
    """Data-parallel training of two processes on one GPU over the host data plane: two epochs of the DP path (gradient
    all-reduce, SyncBN sums, all-rank loss normalisation) beside the single-process epochs over the same global batches."""
    import ctypes

    import numpy as np

    from vamb_amd import _lib, encode as ve, parallel, synth

    _lib.require_gpu()
    comm = parallel.Communicator(comm_control.dist, rccl="host")
    lib = _lib.load()
    n, S, gb = 2048, 6, 256
    ab, tnf, lens, _ = synth.features(n, S, seed=5)
    ve.set_prep_mode("host")
    dl = ve.make_dataloader(ab, tnf, lens, batchsize=gb, destroy=True)
    ve.set_prep_mode("auto")
    tens = [t.numpy() for t in dl.dataset.tensors]
    w_all = tens[3].reshape(-1).astype(np.float64)
    nb = n // gb
    perm = np.random.RandomState(0).permutation(n)[: nb * gb].astype(np.int64).reshape(nb, gb)
    import torch

    def run_serial(rows_per_batch, dataset):
        vae = ve.VAE(S, nhiddens=[64, 48], nlatent=8, dropout=0.0, seed=3)   # no dropout: the masks depend on the rank
        ds = torch.utils.data.TensorDataset(*(torch.from_numpy(np.ascontiguousarray(x)) for x in dataset))
        vae._ensure_dataset(torch.utils.data.DataLoader(ds, batch_size=rows_per_batch.shape[1], shuffle=True, drop_last=True))
        means = (ctypes.c_double * 5)()
        flat = np.ascontiguousarray(rows_per_batch.reshape(-1))
        for _ in range(2):
            _lib.check(lib.vh_vae_train_epoch(vae._h, _lib.ptr(flat), nb, rows_per_batch.shape[1], means))
        return {k: v.numpy().copy() for k, v in vae.state_dict().items()}, list(means)

    # this rank's half of every global batch: rows of its own shard (global rows [rank n / 2, (rank + 1) n / 2))
    half = n // 2
    lo = comm.rank * half
    mine = [np.sort(b[(b >= lo) & (b < lo + half)]) for b in perm]
    k = min(len(m) for m in mine)
    k = int(comm_control.all_reduce_min(np.array([k], np.int64))[0])
    local_rows = np.stack([m[:k] for m in mine]) - lo
    # the serial reference trains on exactly the rows the two ranks use
    used = comm_control.all_gather_arrays(np.stack([m[:k] for m in mine]))
    serial_rows = np.concatenate(used, axis=1)
    shard = [x[lo:lo + half] for x in tens]
    # (the all-rank weight sums must be those of the rows actually used)
    w_used = w_all[serial_rows].sum(axis=1).astype(np.float32)

    def run_dp():
        v = ve.VAE(S, nhiddens=[64, 48], nlatent=8, dropout=0.0, seed=3)
        ds = torch.utils.data.TensorDataset(*(torch.from_numpy(np.ascontiguousarray(x)) for x in shard))
        v._ensure_dataset(torch.utils.data.DataLoader(ds, batch_size=k, shuffle=True, drop_last=True))
        v.attach_communicator(comm)
        means = (ctypes.c_double * 5)()
        flat = np.ascontiguousarray(local_rows.reshape(-1))
        for _ in range(2):
            _lib.check(lib.vh_vae_train_epoch_dp(v._h, _lib.ptr(flat), nb, k, 2 * k, _lib.ptr(w_used), means))
        return {kk: vv.numpy().copy() for kk, vv in v.state_dict().items()}, list(means)

    sd_dp, m_dp = run_dp()
    sd_serial, m_serial = run_serial(serial_rows, tens)
    import hashlib

    digest = hashlib.sha256(b"".join(np.ascontiguousarray(sd_dp[key]).tobytes() for key in sorted(sd_dp))).hexdigest()
    comm.close()
    return {"param_digest": digest, "means_dp": m_dp, "means_serial": m_serial}


def test_two_processes_one_gpu_sharded_stream_equals_reference():
    results = pg._run("test_parallel_gpu:_sharded_cluster_hip", world=2)
    for rank in (0, 1):
        for name, (ok, msg) in results[rank].items():
            assert ok, (rank, name, msg)


@pytest.mark.parametrize("world", [2, 8])
def test_processes_on_one_gpu_native_sharded_state_machine_equals_reference(world):
    """vh_gen_create_sharded / vh_gen_next on 2 and on 8 ranks (BASELINE's 8-GPU partition, uneven shards between 1/22 and 5/22 of
    the rows, all on one device over the host data plane): every rank emits the real reference's golden stream."""
    results = pg._run("test_parallel_gpu:_sharded_cluster_native", world=world)
    for rank in range(world):
        for name, (ok, msg) in results[rank].items():
            assert ok, (rank, name, msg)


def test_two_processes_one_gpu_data_parallel_training():
    """Two ranks, one GPU, the library's host data plane: after two data-parallel epochs (gradient all-reduce, SyncBN sums,
    all-rank loss normalisation) both ranks hold BIT-IDENTICAL parameters and running statistics -- the invariant of the DP
    path -- and the epoch's loss means agree with a single-process run over the same global batches to within the spread of
    the reparameterisation noise (the noise stream is keyed by rank and local row, so the two runs see different epsilons;
    exact equality against a serial oracle with injected noise is tests/test_parallel_gloo.py)."""
    results = pg._run("test_parallel_gpu:_dp_training_host_plane", world=2)
    assert results[0]["param_digest"] == results[1]["param_digest"]
    assert results[0]["means_dp"] == results[1]["means_dp"]
    for a, b in zip(results[0]["means_dp"][:4], results[0]["means_serial"][:4]):
        assert abs(a - b) <= 0.05 * abs(b) + 1e-3, results[0]
