"""GPU test of the data-parallel plumbing with a 1-rank RCCL communicator (the GPU box has one GPU):
ncclCommInitRank / in-stream ncclAllReduce / the flat-gradient path must reproduce the plain
single-GPU epoch exactly (the all-reduce over one rank is the identity and the slab sum order is the
same), and the all-rank loss normalisation must be honoured."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

from vamb_amd import _lib, encode as ve, parallel, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    c = parallel.Communicator(dist, rccl=True)
    yield c
    c.close()
    dist.destroy_process_group()


def _setup(seed=5):
    n, S = 3000, 6
    ab, tnf, lens, _ = synth.features(n, S, seed=seed)
    dl = ve.make_dataloader(ab, tnf, lens, batchsize=256, destroy=True)
    return n, S, dl


def test_rccl_epoch_equals_plain_epoch(comm):
    n, S, dl = _setup()
    lib = _lib.load()
    perm = np.random.RandomState(0).permutation(n)[: (n // 256) * 256].astype(np.int64)
    outs = []
    for use_comm in (False, True):
        vae = ve.VAE(S, nhiddens=[64, 48], nlatent=8, seed=3)
        vae._ensure_dataset(dl)
        if use_comm:
            vae.attach_communicator(comm)
        means = (ctypes.c_double * 5)()
        for _ in range(2):
            if use_comm:
                w = dl.dataset.tensors[3].numpy().reshape(-1).astype(np.float64)
                gw = w[perm].reshape(-1, 256).sum(axis=1).astype(np.float32)
                _lib.check(lib.vh_vae_train_epoch_dp(vae._h, _lib.ptr(perm), n // 256, 256, 256, _lib.ptr(gw), means))
            else:
                _lib.check(lib.vh_vae_train_epoch(vae._h, _lib.ptr(perm), n // 256, 256, means))
        outs.append((vae.state_dict(), list(means), vae.optimizer_state()))
    (sd0, m0, o0), (sd1, m1, o1) = outs
    # dropout masks differ only through the rank key (rank 0 in both) -> identical streams
    for k in sd0:
        a, b = sd0[k].numpy(), sd1[k].numpy()
        assert np.allclose(a, b, rtol=2e-5, atol=1e-7), k
    assert np.allclose(m0, m1, rtol=1e-5)
    assert abs(o0["d"] - o1["d"]) <= 1e-5 * o0["d"]


def test_global_batch_normalisation(comm):
    """Half of a batch with global_batch = 2*local and the global weight sum gives exactly half the
    gradient contribution: loss means scale by 1/2 relative to a self-contained batch with the same
    weight mean."""
    n, S, dl = _setup(seed=6)
    lib = _lib.load()
    rows = np.arange(256, dtype=np.int64)
    w = dl.dataset.tensors[3].numpy().reshape(-1).astype(np.float64)
    res = []
    for gb in (256, 512):
        vae = ve.VAE(S, nhiddens=[64, 48], nlatent=8, dropout=0.0, seed=3)
        vae._ensure_dataset(dl)
        # local BatchNorm statistics: with one rank standing in for half of a global batch the synchronised
        # statistics (sums over all ranks / global batch) would be those of a half-empty batch
        vae.attach_communicator(comm, syncbn=False)
        gw = np.array([w[rows].sum() * (gb // 256)], np.float32)
        means = (ctypes.c_double * 5)()
        _lib.check(lib.vh_vae_train_epoch_dp(vae._h, _lib.ptr(rows), 1, 256, gb, _lib.ptr(gw), means))
        res.append(list(means))
    # the reported local share of the all-rank mean halves when the batch is twice as large
    assert np.allclose(np.array(res[1][1:]), np.array(res[0][1:]) / 2, rtol=1e-3)


@pytest.mark.parametrize("state_machine", ["native", "python"])
@pytest.mark.parametrize("name", ["blob_s008_n2000", "blob_s050_n3000", "blob_s008_n10000"])
def test_rccl_sharded_cluster_stream_matches_golden(comm, name, state_machine, monkeypatch):
    """The row-sharded cluster path on its RCCL data plane with a 1-rank communicator: the stream must equal the reference
    golden, exactly as the single-GPU path does.  "native": vh_gen_create_sharded (the state machine in the library, ONE
    all-gather of accumulators + list parts per pass over RCCL); "python": the Python state machine over
    vh_clu_scan_sharded / vh_clu_select_sharded (query exchange, int64 accumulator all-reduce, select gather)."""
    import hashlib

    if state_machine == "python":
        monkeypatch.setenv("VAMBHIP_PY_GENERATOR", "1")

    import fixture_defs as fd

    mat, lens, kw = fd.cluster_inputs(name)
    gen = parallel.sharded_cluster_generator(comm, mat.copy(), lens, rng_seed=kw.get("rng_seed", 0),
                                             **{k: v for k, v in kw.items() if k != "rng_seed"})
    assert gen._backend.device_plane and gen._sharded_native == (state_machine == "native")
    got = fd.pack_stream(list(gen))
    golden = fd.load("cluster_" + name)
    order_hash = hashlib.sha256(np.argsort(lens)[::-1].astype(np.int64).tobytes()).hexdigest()
    if str(golden["order_sha256"]) != order_hash:
        pytest.skip("np.argsort tie order differs on this CPU (unstable sort, cluster.py:275)")
    ok, msg = fd.streams_equal(got, golden, pvr_rtol=1e-6)
    assert ok, msg


def test_rccl_sharded_scan_equals_local_scan(comm):
    """vh_clu_scan_sharded (k from 1 to 32, incl. the matrix-pipe kernel) == vh_clu_scan on the same rows, bit for bit."""
    from vamb_amd import cluster as vc

    lat, _ = synth.blob_latent(20000, 32, 0.3, seed=4)
    lens = synth.lengths(20000, 4).astype(np.float32)
    a = vc.HipScanBackend(lat.copy(), lens, False, None)
    b = vc.HipScanBackend(lat.copy(), lens, False, None)
    sh = parallel.ShardedScanBackend(comm, b)
    rng = np.random.RandomState(1)
    for k in (1, 3, 8, 9, 16, 25, 32):
        med = rng.choice(20000, k, replace=False).astype(np.int64)
        want = a.scan_raw(med)          # int64 [k, 63] = density_fx, hist_fx[60], n_within, n_lt
        stats = sh.scan(list(med))
        assert np.array_equal(np.stack([st.hist_fx for st in stats]), want[:, 1:61]), k
        assert [st.n_within for st in stats] == want[:, 61].tolist() and [st.n_lt for st in stats] == want[:, 62].tolist(), k
        want_d = (want[:, 0] / _lib.DENSITY_SCALE).astype(np.float32).astype(np.float64)
        assert np.array_equal(np.array([st.density for st in stats]), want_d), k
    rows = sh.select(int(med[0]), 0.2, False)
    assert np.array_equal(rows, a.select(int(med[0]), 0.2, False))
    a.close()
    b.close()
