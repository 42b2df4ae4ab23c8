"""The cluster oracle (oracle/cluster_oracle.py + oracle/cluster_scan.c) against the golden cluster
streams produced by the REAL reference ClusterGenerator (tests/golden/make_golden.py)."""
import numpy as np
import pytest

import fixture_defs as fd
import cluster_oracle as co


def test_edge_table_is_torch_linspace(oracle_lib):
    import torch

    e = np.zeros(61, np.float32)
    oracle_lib.vo_edges(e.ctypes.data)
    assert np.array_equal(e, torch.linspace(0.0, 0.3, 61).numpy())


def test_binning_matches_torch_histogram(oracle_lib):
    import torch

    edges = torch.linspace(0.0, 0.3, 61).numpy()
    rng = np.random.RandomState(0)
    v = np.concatenate([rng.random_sample(3000).astype(np.float32) * 0.32 - 0.01, edges,
                        np.nextafter(edges, np.float32(1)), np.nextafter(edges, np.float32(-1))]).astype(np.float32)
    for x in v:
        h = torch.histogram(torch.tensor([x]), bins=60, range=(0.0, 0.3))[0].numpy()
        nz = np.flatnonzero(h)
        ref = int(nz[0]) if len(nz) else -1
        assert oracle_lib.vo_bin(float(x)) == ref, float(x)


def test_normalize_matches_torch(oracle_lib):
    import torch

    rng = np.random.RandomState(3)
    m = rng.standard_normal((500, 32)).astype(np.float32)
    m[7] = 0
    ours = co.normalize(m.copy())
    t = torch.from_numpy(m.copy())
    zero = (t == 0).all(dim=1)
    t[zero] = 1 / t.shape[1]
    t /= t.norm(dim=1).reshape(-1, 1) * (2 ** 0.5)
    # same formula, different summation order inside norm(): a few ulp
    assert np.abs(ours - t.numpy()).max() < 2e-7
    assert np.allclose((ours.astype(np.float64) ** 2).sum(axis=1), 0.5, atol=1e-6)


def test_smoothing_table_matches_reference_constant():
    # value of vamb/cluster.py:39-73 _NORMALPDF (float32 tensor * python 0.005)
    assert co.NORMALPDF.dtype == np.float32 and len(co.NORMALPDF) == 31
    assert abs(float(co.NORMALPDF.sum()) - 1.0) < 1e-3
    assert co.NORMALPDF[15] == np.float32(0.005) * np.float32(3.98942280e01)


@pytest.mark.parametrize("name", list(fd.CLUSTER_CASES))
def test_oracle_reproduces_reference_stream(oracle_lib, name):
    mat, lens, kw = fd.cluster_inputs(name)
    got = fd.pack_stream(list(co.OracleClusterGenerator(mat.copy(), lens, **kw)))
    want = fd.load("cluster_" + name)
    ok, msg = fd.streams_equal(got, want)
    assert ok, msg


def test_every_point_clustered_once(oracle_lib):
    mat, lens, kw = fd.cluster_inputs("blob_s008_n2000")
    seen = np.zeros(len(mat), int)
    for c in co.OracleClusterGenerator(mat, lens, **kw):
        seen[c.members] += 1
    assert (seen == 1).all()


def test_bad_params(oracle_lib):
    mat, lens, _ = fd.cluster_inputs("blob_zero_dup")
    for kw in (dict(maxsteps=0), dict(windowsize=0), dict(minsuccesses=0), dict(minsuccesses=5, windowsize=4)):
        with pytest.raises(ValueError):
            co.OracleClusterGenerator(mat, lens, **kw)
    with pytest.raises(ValueError):
        co.OracleClusterGenerator(mat.astype(np.float64), lens)
    with pytest.raises(ValueError):
        co.OracleClusterGenerator(mat[:0], lens[:0])


@pytest.mark.parametrize("name", list(fd.CLUSTER_CASES))
def test_reference_order_oracle_stream_equals_the_reference(name):
    """oracle.set_order(1): the evaluation orders measured on the reference's own torch / oneMKL CPU build
    (oracle/probe_reference_order.py) -- `matmul` and `norm` (equal to torch bit for bit for every latent width), and the two
    float32 sums the reference reports: torch.sum of the density terms and torch.histogram's one-thread bin sums.  The restated
    state machine then reproduces the real reference's golden streams EXACTLY, every field, observed_pvr included (the two
    100 k fixtures take minutes on the CPU: oracle/check_reference_order_streams.py ->
    profiles/r03_reference_order_streams.txt; on the GPU: tests/test_cluster_gpu.py::test_reference_order_*)."""
    co.set_order(1)
    try:
        mat, lens, kw = fd.cluster_inputs(name)
        got = fd.pack_stream(list(co.OracleClusterGenerator(mat.copy(), lens, **kw)))
    finally:
        co.set_order(co.DEFAULT_ORDER)
    golden = fd.load("cluster_" + name)
    ok, msg = fd.streams_equal(got, golden, pvr_rtol=0.0)
    assert ok, msg


def test_reference_sums_equal_torch_bit_for_bit():
    """vo_torch_sum == torch.sum and vo_reference_sums' histogram == torch.histogram (one thread), bit for bit."""
    import torch

    if torch.backends.cpu.get_cpu_capability() != "AVX512":
        pytest.skip("measured on an AVX-512 host")
    torch.set_num_threads(1)
    rng = np.random.RandomState(3)
    for n in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 511, 512, 1000, 1024, 2047, 4096, 9000, 33000]:
        for _ in range(8):
            x = (rng.standard_normal(n) * rng.uniform(0.1, 100)).astype(np.float32)
            assert co.torch_sum(x) == float(torch.from_numpy(x).sum().item()), n
    for n in (10, 1000, 20000):
        d = rng.uniform(0, 0.35, n).astype(np.float32)
        w = rng.randint(2000, 100000, n).astype(np.float32)
        sel = d <= np.float32(0.3)
        hist, edges = torch.zeros(60), torch.zeros(61)
        torch.histogram(input=torch.from_numpy(d[sel]), bins=60, range=(0.0, 0.3), out=(hist, edges), weight=torch.from_numpy(w[sel]))
        dens, h = co.reference_sums(d, w, None)
        assert np.array_equal(h.view(np.uint32), hist.numpy().view(np.uint32))
        close = (np.float32(0.05) - d[d <= np.float32(0.05)]).astype(np.float32)
        want = float((torch.from_numpy(w[d <= np.float32(0.05)]) * torch.from_numpy(close)).sum().item())
        assert dens == want


def test_reference_order_equals_torch_bit_for_bit():
    """The pin of oracle.set_order(1): on an AVX-512 host the restated order IS what torch computes for the reference's two
    reductions -- `matrix.matmul(matrix[index])` (cluster.py:674) and `matrix.norm(dim=1)` in `_normalize` (cluster.py:668) --
    for the latent widths of every fixture and of BASELINE's configs (32, 64) and a few awkward ones."""
    import torch

    if torch.backends.cpu.get_cpu_capability() != "AVX512":
        pytest.skip("the measured order is that of torch / oneMKL on an AVX-512 host")
    rng = np.random.RandomState(11)
    co.set_order(1)
    try:
        for threads in (1, 4):
            torch.set_num_threads(threads)
            for L in (1, 3, 12, 16, 17, 29, 32, 33, 40, 48, 49, 64, 65):
                n = 6000
                raw = (rng.standard_normal((n, L)) * rng.uniform(0.1, 3, (n, 1))).astype(np.float32)
                raw[5] = 0
                t = torch.from_numpy(raw.copy())
                zero = (t == 0).all(dim=1)
                t[zero] = 1 / L
                t /= t.norm(dim=1).reshape(-1, 1) * (2 ** 0.5)                      # cluster.py:664-668
                ours = co.normalize(raw.copy())
                assert np.array_equal(ours.view(np.uint32), t.numpy().view(np.uint32)), (threads, L)
                for idx in (0, 5, n // 2, n - 1):
                    want = (0.5 - t.matmul(t[idx])).numpy()                          # cluster.py:674
                    want[idx] = 0.0
                    got = co.scan(ours, np.ones(n, np.float32), None, idx)["dist"]
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (threads, L, idx)
    finally:
        co.set_order(co.DEFAULT_ORDER)
        torch.set_num_threads(1)
