"""Host logic of vamb_amd.cluster.ClusterGenerator (seed walk, speculative wander, threshold walk,
lazy packing) against the REAL reference's golden cluster streams, with the device passes replaced
by the oracle backend (tests/oracle_backend.py).  Runs without a GPU."""
import numpy as np
import pytest

import fixture_defs as fd
from oracle_backend import OracleScanBackend
from vamb_amd import cluster as vc


@pytest.mark.parametrize("name", list(fd.CLUSTER_CASES))
@pytest.mark.parametrize("pack_min", [16, 10 ** 9])
def test_stream_matches_reference(oracle_lib, monkeypatch, name, pack_min):
    monkeypatch.setattr(vc.ClusterGenerator, "PACK_MIN_ROWS", pack_min)
    mat, lens, kw = fd.cluster_inputs(name)
    gen = vc.ClusterGenerator(mat.copy(), lens, _backend_factory=OracleScanBackend, **kw)
    got = fd.pack_stream(list(gen))
    ok, msg = fd.streams_equal(got, fd.load("cluster_" + name))
    assert ok, msg


def test_speculation_batches_candidates(oracle_lib):
    mat, lens, kw = fd.cluster_inputs("blob_s008_n2000")
    gen = vc.ClusterGenerator(mat.copy(), lens, _backend_factory=OracleScanBackend, **kw)
    list(gen)
    b = gen._backend
    assert b.max_batch > 1 and b.scan_medoids > b.scan_passes


def test_smoothing_and_threshold_match_oracle(oracle_lib):
    import cluster_oracle as co

    rng = np.random.RandomState(0)
    assert np.array_equal(vc._NORMALPDF, co.NORMALPDF)
    for _ in range(200):
        hist = (rng.random_sample(60) * rng.choice([0, 1, 5e4], 60)).astype(np.float32)
        a, b = vc.smooth_histogram(hist), co.smooth(hist)
        assert np.array_equal(a, b)
        for pvr in (0.1, 0.30000000000000004, 0.6):
            t1 = vc.threshold_from_densities(a, pvr)
            t2 = co.pick_threshold(b, pvr)
            assert (t2 is None and isinstance(t1, vc.NoThreshold)) or t1 == t2


def test_bad_params_and_protocol(oracle_lib):
    mat, lens, _ = fd.cluster_inputs("blob_zero_dup")
    mk = lambda *a, **k: vc.ClusterGenerator(*a, _backend_factory=OracleScanBackend, **k)  # noqa: E731
    with pytest.raises(ValueError):
        mk(mat.astype(np.float64), lens)
    for kw in (dict(maxsteps=0), dict(windowsize=0), dict(minsuccesses=0), dict(minsuccesses=5, windowsize=4)):
        with pytest.raises(ValueError):
            mk(mat, lens, **kw)
    with pytest.raises(ValueError):
        mk(np.zeros((0, 40), np.float32), np.array([], dtype=int))
    with pytest.raises(ValueError):
        mk(mat, lens[:-1])
    g = mk(mat.copy(), lens)
    assert iter(g) is g
    c = next(g)
    assert isinstance(c, vc.Cluster) and isinstance(c.members, np.ndarray)


def test_destroy_and_normalized_semantics(oracle_lib):
    # reference test/test_cluster.py:57-87
    mat, lens, _ = fd.cluster_inputs("test_cluster_py")
    mk = lambda *a, **k: vc.ClusterGenerator(*a, _backend_factory=OracleScanBackend, **k)  # noqa: E731
    before = mat.copy()
    g = mk(mat, lens)
    assert np.array_equal(before, mat)
    assert np.any(np.abs(mat - g.matrix.numpy()) > 0.001)
    cp = mat.copy()
    g = mk(cp, lens, destroy=True)
    assert np.all(np.abs(cp - g.matrix.numpy()) < 1e-6)
    assert np.any(np.abs(mat - cp) > 0.001)
    cp2 = cp.copy()
    mk(cp2, lens, destroy=True, normalized=True)
    assert np.array_equal(cp, cp2)
