"""Multi-GPU execution of the hot path: one process per GPU, contigs sharded by row.

The reference has no distributed path (no NCCL/MPI call site), so nothing here
mirrors reference code; it is the host side of BASELINE.json's north star ("training mini-batches and
the cluster seed search partition across the 8 GPUs ... RCCL all-reduce of gradients and all-gather of
per-shard neighbour histograms").

Two planes
  * bulk data plane: RCCL called from inside libvambhip on the library's own HIP stream
    (``vh_comm_*`` / ``vh_vae_train_epoch_dp``): the flat gradient all-reduce of every optimisation
    step, the per-epoch BatchNorm-buffer and loss reductions.  No host synchronisation per step.
  * control plane: a ``torch.distributed`` process group (gloo is enough: a few hundred bytes per
    message) for bootstrap (the ncclUniqueId), per-epoch batch planning and the cluster scan's
    scalar/histogram reductions and candidate-list gathers.

Training semantics under data parallelism: every rank holds a contiguous row shard and contributes
``global_batch / world`` rows to each step; the loss of a step is normalised by the ALL-RANK batch so
the summed gradients equal the single-GPU gradient of the same global batch.  BatchNorm batch
statistics are synchronised by default (``syncbn=True``: the fp64 sums of every BatchNorm layer are
all-reduced in stream order, forward and backward, so a step computes what the single-process
reference computes on the global batch, vamb/encode.py:238,246); ``syncbn=False`` keeps per-rank
statistics (torch DDP without SyncBatchNorm) and averages the running statistics over the ranks after
every epoch so that all ranks encode with the same network.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as _np

from . import _lib
from .cluster import _MAX_MEDOIDS_PER_PASS, ScanStats


class Communicator:
    """A torch.distributed group (control plane) plus, optionally, an RCCL communicator owned by
    libvambhip (data plane)."""

    def __init__(self, dist, group=None, rccl=True):
        """``rccl``: True = an RCCL communicator inside libvambhip (one rank per GPU); ``"host"`` = the library's HOST data
        plane: its collectives are carried out by this process group through two callbacks (``vh_comm_create_host``) --
        for running the multi-rank device paths with several processes on one GPU, where RCCL refuses to start; False = no
        device data plane at all (the exchanges of the sharded scan run in Python on the control plane)."""
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.handle = None
        self._lib = None
        self.data_plane = "rccl" if rccl is True else ("host" if rccl == "host" else None)
        if rccl == "host":
            self._lib = _lib.load()
            _lib.sync_env_options()
            self._make_host_plane()
        elif rccl:
            self._lib = _lib.load()
            _lib.sync_env_options()
            ident = (ctypes.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(self._lib.vh_comm_unique_id(ident))
            payload = [bytes(ident)]
            dist.broadcast_object_list(payload, src=0, group=group)
            ident = (ctypes.c_ubyte * 128).from_buffer_copy(payload[0])
            h = ctypes.c_void_p()
            _lib.check(self._lib.vh_comm_create(self.rank, self.world, ident, ctypes.byref(h)))
            self.handle = h

    @classmethod
    def from_torch_distributed(cls, dist, group=None, rccl=True) -> "Communicator":
        return cls(dist, group, rccl)

    def _make_host_plane(self):
        import traceback

        import torch

        dist, group, world = self.dist, self.group, self.world
        np_types = {0: (_np.float32, ctypes.c_float), 1: (_np.float64, ctypes.c_double), 2: (_np.int64, ctypes.c_int64)}

        def allreduce(_ctx, buf, count, dtype):
            try:
                _, ct = np_types[int(dtype)]   # (uint64 sums as int64: the same bits)
                arr = _np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ct)), shape=(int(count),))
                dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:   # an exception must not unwind through the C caller
                traceback.print_exc()
                return 1

        def allgather(_ctx, send, recv, nbytes):
            try:
                nbytes = int(nbytes)
                s_arr = _np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
                r_arr = _np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes * world,))
                parts = [torch.from_numpy(r_arr[i * nbytes:(i + 1) * nbytes]) for i in range(world)]
                dist.all_gather(parts, torch.from_numpy(s_arr.copy()), group=group)
                return 0
            except Exception:
                traceback.print_exc()
                return 1

        ar_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)
        ag_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
        self._callbacks = (ar_t(allreduce), ag_t(allgather))   # kept alive as long as the communicator
        h = ctypes.c_void_p()
        _lib.check(self._lib.vh_comm_create_host(self.rank, self.world, ctypes.cast(self._callbacks[0], ctypes.c_void_p),
                                                 ctypes.cast(self._callbacks[1], ctypes.c_void_p), None, ctypes.byref(h)))
        self.handle = h

    def info(self) -> dict:
        """rank / world, the rank count the collective library itself reports (ncclCommCount) and the kind of data plane."""
        if self.handle is None:
            return {"rank": self.rank, "world": self.world, "reported_ranks": None, "data_plane": None}
        r, w, n, is_rccl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self._lib.vh_comm_info(self.handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(n), ctypes.byref(is_rccl)))
        return {"rank": r.value, "world": w.value, "reported_ranks": n.value, "data_plane": "rccl" if is_rccl.value else "host"}

    def close(self):
        if self.handle is not None and self._lib is not None:
            self._lib.vh_comm_destroy(self.handle)
            self.handle = None

    # ---- control-plane helpers (numpy in, numpy out) ------------------------------------------
    def _tensor(self, a: _np.ndarray):
        import torch

        return torch.from_numpy(a)

    def all_reduce_sum(self, a: _np.ndarray) -> _np.ndarray:
        a = _np.ascontiguousarray(a).copy()
        self.dist.all_reduce(self._tensor(a), op=self.dist.ReduceOp.SUM, group=self.group)
        return a

    def all_reduce_min(self, a: _np.ndarray) -> _np.ndarray:
        a = _np.ascontiguousarray(a).copy()
        self.dist.all_reduce(self._tensor(a), op=self.dist.ReduceOp.MIN, group=self.group)
        return a

    def all_gather_arrays(self, a: _np.ndarray) -> list:
        """Variable-length gather of 1-d / 2-d arrays (same dtype and trailing shape on every rank)."""
        out = [None] * self.world
        self.dist.all_gather_object(out, _np.ascontiguousarray(a), group=self.group)
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)


# -------------------------------------------------------------------------------------------------
# training: per-epoch plan of a data-parallel epoch
# -------------------------------------------------------------------------------------------------
def plan_epoch(comm: Communicator, n_local: int, global_batch: int, weights_local: _np.ndarray, perm: _np.ndarray):
    """Split one epoch over the ranks.

    perm is this rank's permutation of its LOCAL rows.  Returns (rows, n_batches, local_batch,
    global_wsum): rows = perm truncated to n_batches*local_batch, n_batches = the minimum over ranks
    of full local batches (every rank must run the same number of collective steps), and
    global_wsum[b] = sum over ALL ranks of the weights of batch b (the loss normaliser).
    """
    if global_batch % comm.world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by the number of GPUs {comm.world}")
    local_batch = global_batch // comm.world
    if local_batch < 2:
        raise ValueError("every GPU needs at least 2 rows per step (BatchNorm)")
    if n_local < local_batch:
        raise ValueError(f"shard of {n_local} rows is smaller than the per-GPU batch {local_batch}")
    n_batches = int(comm.all_reduce_min(_np.array([n_local // local_batch], dtype=_np.int64))[0])
    rows = _np.ascontiguousarray(perm[: n_batches * local_batch], dtype=_np.int64)
    w = _np.asarray(weights_local, dtype=_np.float64).reshape(-1)
    local_wsum = w[rows].reshape(n_batches, local_batch).sum(axis=1)
    global_wsum = comm.all_reduce_sum(local_wsum).astype(_np.float32)
    return rows, n_batches, local_batch, global_wsum


# -------------------------------------------------------------------------------------------------
# clustering: the scan backend over row shards
# -------------------------------------------------------------------------------------------------
class ShardedScanBackend:
    """Scan backend for ``ClusterGenerator`` whose matrix is row-sharded over the ranks of ``comm``.

    Rank r holds the contiguous global rows [offset[r], offset[r+1]).  Every rank runs the same host
    logic (same RNG seed) and calls the same methods in the same order; each call is
      local device pass  +  one small reduction / gather on the control plane.
    Integer accumulators make the reduced results independent of the number of shards: the cluster
    stream is bit-identical to the single-GPU one.
    """

    def __init__(self, comm: Communicator, local_backend):
        self.comm = comm
        self.local = local_backend
        self.L = local_backend.L
        # Device data plane: with an RCCL communicator and the HIP backend, the query-vector exchange, the accumulator
        # all-reduce and the select gather run inside libvambhip on the scan stream (vh_clu_scan_sharded /
        # vh_clu_select_sharded).  Otherwise (CPU tests with the oracle backend, rccl=False) the same exchanges go
        # through the torch.distributed control plane below; integer accumulators make both bit-identical.
        self.device_plane = getattr(comm, "handle", None) is not None and getattr(local_backend, "h", None) is not None \
            and hasattr(local_backend, "lib")
        if self.device_plane:
            _lib.check(local_backend.lib.vh_clu_attach_comm(local_backend.h, comm.handle))
            self._res = (_lib.ScanResult * _MAX_MEDOIDS_PER_PASS)()
            self._sel = _np.empty(1, _np.int64)
            self.lib, self.h = local_backend.lib, local_backend.h   # (the native sharded state machine drives the shard's handle)
        self._refresh_offsets()
        self.scan_passes = 0
        self.scan_medoids = 0
        self.rows_streamed = 0
        self.kernel_ms = 0.0

    def _refresh_offsets(self):
        counts = _np.zeros(self.comm.world, dtype=_np.int64)
        counts[self.comm.rank] = self.local.n_rows
        counts = self.comm.all_reduce_sum(counts)
        self.offsets = _np.ascontiguousarray(_np.concatenate([[0], _np.cumsum(counts)]), dtype=_np.int64)
        self.n_rows = int(self.offsets[-1])
        if getattr(self, "device_plane", False) and len(self._sel) < self.n_rows:
            self._sel = _np.empty(max(1, self.n_rows), _np.int64)

    def _owner_local(self, rows):
        """global rows -> (local row or -1 per entry) for this rank"""
        rows = _np.asarray(rows, dtype=_np.int64)
        lo, hi = self.offsets[self.comm.rank], self.offsets[self.comm.rank + 1]
        mine = (rows >= lo) & (rows < hi)
        return _np.where(mine, rows - lo, -1), mine

    def _queries(self, rows):
        """[k, L] query vectors of global rows: owners contribute their rows, the rest zeros; a sum
        all-reduce distributes them exactly (x + 0 == x)."""
        local_rows, mine = self._owner_local(rows)
        q = _np.zeros((len(local_rows), self.L), dtype=_np.float32)
        if mine.any():
            q[mine] = self.local.get_rows(local_rows[mine])
        return self.comm.all_reduce_sum(q), local_rows

    def scan(self, medoids):
        out = []
        step = getattr(self.local, "max_medoids", _MAX_MEDOIDS_PER_PASS)
        for lo in range(0, len(medoids), step):
            chunk = medoids[lo:lo + step]
            if self.device_plane:
                local_rows, _ = self._owner_local(chunk)
                local_rows = _np.ascontiguousarray(local_rows, dtype=_np.int64)
                k = len(local_rows)
                _lib.check(self.local.lib.vh_clu_scan_sharded(self.local.h, k, _lib.ptr(local_rows), self._res))
                raw = _np.frombuffer(self._res, dtype=_np.int64, count=k * 63).reshape(k, 63).copy()
            else:
                q, local_rows = self._queries(chunk)
                raw = self.local.scan_raw(local_rows, q)          # int64 [k, 63]
                raw = self.comm.all_reduce_sum(raw)
            self.scan_passes += 1
            self.scan_medoids += len(chunk)
            self.rows_streamed += self.local.n_rows
            out.extend(ScanStats.batch(_np.asarray(raw, dtype=_np.int64)))
        return out

    def select(self, medoid: int, threshold: float, remove: bool) -> _np.ndarray:
        if self.device_plane:
            local_rows, _ = self._owner_local([medoid])
            n = ctypes.c_int64(0)
            thr = float(_np.float32(threshold))
            _lib.check(self.local.lib.vh_clu_select_sharded(self.local.h, int(local_rows[0]), thr, int(remove),
                                                            _lib.ptr(self.offsets), _lib.ptr(self._sel), len(self._sel),
                                                            ctypes.byref(n)))
            self.scan_passes += 1
            self.rows_streamed += self.local.n_rows
            return self._sel[: n.value].copy()
        q, local_rows = self._queries([medoid])
        rows = self.local.select_query(int(local_rows[0]), q[0], threshold, remove)
        self.scan_passes += 1
        self.rows_streamed += self.local.n_rows
        parts = self.comm.all_gather_arrays(rows + self.offsets[self.comm.rank])
        return _np.concatenate(parts).astype(_np.int64)

    def remove(self, rows):
        local_rows, mine = self._owner_local(rows)
        if mine.any():
            self.local.remove(local_rows[mine])

    def pack(self) -> int:
        self.local.pack()
        self._refresh_offsets()
        return self.n_rows

    def matrix(self) -> _np.ndarray:
        return _np.concatenate(self.comm.all_gather_arrays(self.local.matrix()), axis=0)

    def set_timing(self, on: bool):
        if hasattr(self.local, "set_timing"):
            self.local.set_timing(on)

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()


def sharded_cluster_generator(comm: Communicator, local_matrix: _np.ndarray, local_lengths: _np.ndarray,
                              maxsteps: int = 25, windowsize: int = 300, minsuccesses: int = 15,
                              destroy: bool = False, normalized: bool = False, rng_seed: int = 0,
                              _local_backend_factory=None):
    """``ClusterGenerator`` over a row-sharded latent matrix.  Every rank passes its own shard (rank
    order == global row order) and iterates the returned generator in lock step; all ranks receive the
    same stream of clusters with GLOBAL row indices."""
    from .cluster import ClusterGenerator, HipScanBackend

    local_matrix = _np.ascontiguousarray(local_matrix)
    if local_matrix.dtype != _np.float32:
        raise ValueError("Matrix must be of dtype float32")
    if len(local_lengths) != len(local_matrix):
        raise ValueError("N sequences in lengths and matrix do not match")
    lengths_global = _np.concatenate(comm.all_gather_arrays(_np.asarray(local_lengths)))
    lf = _np.ascontiguousarray(local_lengths, dtype=_np.float32)
    factory = HipScanBackend if _local_backend_factory is None else _local_backend_factory
    out = local_matrix if (destroy and not normalized) else None
    local = factory(local_matrix, lf, normalized, out)
    backend = ShardedScanBackend(comm, local)
    # HIP shards + a device data plane (RCCL, or the host plane of the tests): the NATIVE sharded state machine
    # (vh_gen_create_sharded: same speculation and lazy validation as on one GPU, one collective per pass, no Python between
    # the passes); otherwise the Python state machine on this backend (the CPU oracle backend of the tests, rccl=False)
    return ClusterGenerator.from_backend(backend, lengths_global, maxsteps=maxsteps, windowsize=windowsize,
                                         minsuccesses=minsuccesses, rng_seed=rng_seed)
