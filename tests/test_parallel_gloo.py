"""world_size 2 and 8 CPU tests (gloo) of the multi-GPU host path in vamb_amd/parallel.py.

The device passes are replaced by the oracle backend (tests/oracle_backend.py) and the per-rank VAE
step by the numpy oracle, so these tests exercise exactly the product's sharding / reduction logic:
  * ShardedScanBackend + ClusterGenerator.from_backend: the cluster stream produced by 2 row shards
    is bit-identical to the reference's golden single-process stream;
  * plan_epoch: the per-epoch split (equal step counts, global weight sums);
  * the data-parallel loss normalisation: shard gradients summed over ranks equal the gradient of the
    all-rank batch computed serially.
"""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn_name, queue):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        from vamb_amd import parallel

        comm = parallel.Communicator(dist, rccl=False)
        if ":" in fn_name:   # "module:function" -- a per-rank body defined in another test module
            import importlib

            mod, fn = fn_name.split(":")
            result = getattr(importlib.import_module(mod), fn)(comm)
        else:
            result = globals()[fn_name](comm)
        dist.barrier()
        dist.destroy_process_group()
        queue.put((rank, "ok", result))
    except Exception:
        queue.put((rank, "error", traceback.format_exc()))


def _run(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, status, payload = q.get(timeout=600)
        assert status == "ok", f"rank {rank} failed:\n{payload}"
        results[rank] = payload
    for p in procs:
        p.join(timeout=60)
    return results


def uneven_cuts(n, world):
    """Row boundaries of `world` UNEVEN contiguous shards of n rows (world = 2: the 37 % / 63 % split of rounds 1-4)."""
    if world == 2:
        return [0, int(n * 0.37), n]
    w = np.array([(1 + (3 * r) % 5) for r in range(world)], dtype=np.float64)   # 1, 4, 2, 5, 3, 1, 4, 2: shares between 1/22 and 5/22
    cuts = [0] + [int(round(x)) for x in np.cumsum(w) / w.sum() * n]
    cuts[-1] = n
    assert all(b > a for a, b in zip(cuts, cuts[1:]))
    return cuts


# ---- per-rank bodies (module level so that spawn can pickle them by name) -------------------------
def _sharded_cluster(comm):
    import fixture_defs as fd
    from oracle_backend import OracleScanBackend
    from vamb_amd import cluster as vc, parallel

    vc.ClusterGenerator.PACK_MIN_ROWS = 64   # make the lazy packing happen on small fixtures
    out = {}
    for name in ("blob_s008_n2000", "blob_s050_window", "blob_zero_dup"):
        mat, lens, kw = fd.cluster_inputs(name)
        cut = uneven_cuts(len(mat), comm.world)
        lo, hi = cut[comm.rank], cut[comm.rank + 1]
        gen = parallel.sharded_cluster_generator(comm, mat[lo:hi].copy(), lens[lo:hi],
                                                 _local_backend_factory=OracleScanBackend, **kw)
        got = fd.pack_stream(list(gen))
        ok, msg = fd.streams_equal(got, fd.load("cluster_" + name))
        out[name] = (ok, msg)
    return out


def _plan_and_grads(comm):
    import fixture_defs as fd
    import vae_oracle as vo
    from vamb_amd import parallel

    name = "vae_small_nodrop"
    c = fd.VAE_CASES[name]
    g = fd.load(name)
    n = 36                                         # 18 rows per rank
    d, t, a, w = (g[k][:n] for k in ("depths", "tnf", "total_abundance", "weights"))
    lo, hi = comm.rank * 18, (comm.rank + 1) * 18
    perm = np.random.RandomState(comm.rank).permutation(18)
    rows, n_batches, local_batch, gwsum = parallel.plan_epoch(comm, 18, 12, w[lo:hi], perm)
    assert (n_batches, local_batch) == (3, 6) and len(rows) == 18 and gwsum.shape == (3,)
    # batch 0 of every rank: local rows -> global rows
    gb = np.concatenate(comm.all_gather_arrays(rows[:6] + lo))
    st0 = vo.init_state(c["nsamples"], c["nhiddens"], c["nlatent"], c["seed"])
    eps = np.zeros((6, c["nlatent"]), np.float32)

    def shard_grads(global_rows):
        m = vo.OracleVAE(c["nsamples"], c["nhiddens"], c["nlatent"], c["alpha"], c["beta"], 0.0, state=st0)
        do, to, ao, mu = m.forward(d[global_rows], t[global_rows], a[global_rows], eps=eps, masks=None, train=True)
        m.calc_loss(d[global_rows], do, t[global_rows], to, a[global_rows], ao, mu, w[global_rows],
                    global_wsum=float(gwsum[0]), global_batch=12)
        return m.backward()

    mine = shard_grads(rows[:6] + lo)
    flat = np.concatenate([mine[k].reshape(-1) for k in sorted(mine)])
    summed = comm.all_reduce_sum(flat)
    # serial emulation of both shards on every rank
    serial = None
    for r in range(comm.world):
        gr = shard_grads(gb[r * 6:(r + 1) * 6])
        f = np.concatenate([gr[k].reshape(-1) for k in sorted(gr)])
        serial = f if serial is None else serial + f
    err = float(np.abs(summed - serial).max() / np.abs(serial).max())
    wsum_check = float(abs(gwsum[0] - w[gb].sum()))
    return dict(err=err, wsum_err=wsum_check)


def _syncbn_grads(comm):
    """Data parallelism with synchronised BatchNorm: every rank runs forward / backward on ITS half of a batch, the
    BatchNorm batch sums (forward: sum h, sum h^2; backward: sum dy, sum dy xhat) are all-reduced, the loss is
    normalised by the all-rank batch -- the summed gradients and the running statistics must equal the serial
    single-process step on the whole batch (the reference's semantics)."""
    import fixture_defs as fd
    import vae_oracle as vo

    name = "vae_small_drop"
    c = fd.VAE_CASES[name]
    g = fd.load(name)
    masks, eps = fd.vae_randomness(name)
    B = c["batch"]                     # 24 rows: 12 per rank
    half = B // comm.world
    lo, hi = comm.rank * half, (comm.rank + 1) * half
    d, t, a, w = (g[k][:B] for k in ("depths", "tnf", "total_abundance", "weights"))
    st0 = vo.init_state(c["nsamples"], c["nhiddens"], c["nlatent"], c["seed"])
    m = vo.OracleVAE(c["nsamples"], c["nhiddens"], c["nlatent"], c["alpha"], c["beta"], c["dropout"], state=st0,
                     bn_sync=lambda x: comm.all_reduce_sum(np.asarray(x, dtype=np.float64)))
    mk = [mm[lo:hi] for mm in masks[0]]
    do, to, ao, mu = m.forward(d[lo:hi], t[lo:hi], a[lo:hi], eps=eps[0][lo:hi], masks=mk, train=True)
    m.calc_loss(d[lo:hi], do, t[lo:hi], to, a[lo:hi], ao, mu, w[lo:hi], global_wsum=float(w.sum()), global_batch=B)
    mine = m.backward()
    keys = sorted(mine)
    summed = comm.all_reduce_sum(np.concatenate([mine[k].reshape(-1) for k in keys]))
    # the serial truth: one process, the whole batch
    s = vo.OracleVAE(c["nsamples"], c["nhiddens"], c["nlatent"], c["alpha"], c["beta"], c["dropout"], state=st0)
    do, to, ao, mu = s.forward(d, t, a, eps=eps[0], masks=masks[0], train=True)
    s.calc_loss(d, do, t, to, a, ao, mu, w)
    ref = s.backward()
    flat_ref = np.concatenate([ref[k].reshape(-1) for k in keys])
    err = float(np.abs(summed - flat_ref).max() / np.abs(flat_ref).max())
    rs_err = max(float(np.abs(m.state[k] - s.state[k]).max()) for k in m.state if "running" in k)
    return dict(err=err, running_err=rs_err)


def test_syncbn_data_parallel_equals_serial_global_batch(oracle_lib):
    res = _run("_syncbn_grads")
    for rank, out in res.items():
        assert out["err"] < 1e-7 and out["running_err"] < 1e-12, (rank, out)   # E[h^2] - mean^2 vs mean((h - mean)^2) in fp64


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_cluster_stream_is_identical(oracle_lib, world):
    """2 and 8 uneven row shards (BASELINE's 8-GPU partition; shard sizes between 1/22 and 5/22 of the rows): every rank emits the
    reference's golden single-process stream."""
    res = _run("_sharded_cluster", world=world)
    assert sorted(res) == list(range(world))
    for rank, out in res.items():
        for name, (ok, msg) in out.items():
            assert ok, f"rank {rank} {name}: {msg}"


def test_data_parallel_plan_and_gradient_sum(oracle_lib):
    res = _run("_plan_and_grads")
    for rank, out in res.items():
        assert out["err"] < 1e-12 and out["wsum_err"] < 1e-4, (rank, out)


def test_plan_epoch_validation():
    from vamb_amd import parallel

    class FakeComm:
        world, rank = 4, 0

        def all_reduce_min(self, a):
            return a

        def all_reduce_sum(self, a):
            return a

    with pytest.raises(ValueError):
        parallel.plan_epoch(FakeComm(), 100, 10, np.ones(100), np.arange(100))     # 10 % 4 != 0
    with pytest.raises(ValueError):
        parallel.plan_epoch(FakeComm(), 100, 4, np.ones(100), np.arange(100))      # 1 row per GPU
    with pytest.raises(ValueError):
        parallel.plan_epoch(FakeComm(), 3, 16, np.ones(3), np.arange(3))           # shard < local batch
    rows, nb, lb, ws = parallel.plan_epoch(FakeComm(), 100, 16, np.full(100, 2.0), np.arange(100)[::-1].copy())
    assert (nb, lb) == (25, 4) and len(rows) == 100 and np.allclose(ws, 8.0)
