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
        cut = [0, int(len(mat) * 0.37), len(mat)]       # uneven shards
        lo, hi = cut[comm.rank], cut[comm.rank + 1]
        gen = parallel.sharded_cluster_generator(comm, mat[lo:hi].copy(), lens[lo:hi], **kw)   # default factory: HipScanBackend
        assert type(gen._backend.local).__name__ == "HipScanBackend"
        got = fd.pack_stream(list(gen))
        out[name] = fd.streams_equal(got, fd.load("cluster_" + name))
        gen._backend.local.close()
    return out


def test_two_processes_one_gpu_sharded_stream_equals_reference():
    results = pg._run("test_parallel_gpu:_sharded_cluster_hip", world=2)
    for rank in (0, 1):
        for name, (ok, msg) in results[rank].items():
            assert ok, (rank, name, msg)
