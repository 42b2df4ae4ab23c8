"""Iterative medoid clustering on MI355X -- drop-in for ``vamb.cluster``.

``ClusterGenerator`` keeps the reference's constructor signature, iterator protocol, ``Cluster``
objects and ``ValueError`` behaviour (``/root/reference/vamb/cluster.py:122-292``), so
``vamb.__main__.cluster_and_write_files`` (``__main__.py:1277-1289``) can use it unchanged.

Division of labour
  * host (this file): the sequential decision logic, kept literally equivalent to the reference --
    length-ordered seed walk (cluster.py:342-384), ``random.Random`` candidate sampling
    (415-450), 31-tap smoothing + peak/valley walk (483-543), success window / PVR relaxation
    (386-413), cluster emission (545-604).
  * device (``csrc/cluster.hip`` through the C ABI): every pass over the latent matrix -- distance
    scan with density / loner count / length-weighted histogram (606-637, 452-481), threshold
    selection (640-650), row removal (304-309) and order-preserving compaction (318-335).

Differences from the reference that do not change results
  * the <= ``maxsteps`` candidates of one ``wander_medoid`` round are scanned speculatively in ONE
    pass over the matrix (``sample_medoid`` is a pure, cached function in the reference, so looking
    ahead is unobservable); the RNG stream and the order in which candidates are judged are identical.
  * rows of emitted clusters are masked on the device and physically compacted only when fewer than
    half of the resident rows are live (the reference's CPU path compacts after every cluster); all
    decisions use live rows only, i.e. the CPU-path semantics (hard part 5).
  * distances are never materialised: the scan returns exact integer accumulators.
"""
from __future__ import annotations

import ctypes
import os as _os
import random as _random
from collections import deque as _deque
from typing import Optional

import numpy as _np

from . import _lib

_DEFAULT_RADIUS = 0.06
_MEDOID_RADIUS = 0.05
_DELTA_X = 0.005
_XMAX = 0.3
_NBINS = 60
_MAX_MEDOIDS_PER_PASS = 32

# N(0, 0.01) pdf on [-0.075, 0.075] in steps of 0.005, scaled by 0.005 in float32 -- the constant of
# vamb/cluster.py:39-73 (float32 tensor times python float => float32 * float32).
_NORMALPDF = (_np.float32(_DELTA_X) * _np.array(
    [2.43432053e-11, 9.13472041e-10, 2.66955661e-08, 6.07588285e-07, 1.07697600e-05, 1.48671951e-04,
     1.59837411e-03, 1.33830226e-02, 8.72682695e-02, 4.43184841e-01, 1.75283005e00, 5.39909665e00,
     1.29517596e01, 2.41970725e01, 3.52065327e01, 3.98942280e01, 3.52065327e01, 2.41970725e01,
     1.29517596e01, 5.39909665e00, 1.75283005e00, 4.43184841e-01, 8.72682695e-02, 1.33830226e-02,
     1.59837411e-03, 1.48671951e-04, 1.07697600e-05, 6.07588285e-07, 2.66955661e-08, 9.13472041e-10,
     2.43432053e-11], dtype=_np.float32)).astype(_np.float32)


class Loner:
    __slots__ = []


class NoThreshold:
    __slots__ = []


class Cluster:
    """Same attribute surface as the reference's ``Cluster`` (cluster.py:76-119)."""

    __slots__ = ["medoid", "seed", "members", "maximal_pvr", "observed_pvr", "radius", "isdefault",
                 "successes", "attempts"]

    def __init__(self, medoid: int, seed: int, members: _np.ndarray, maximal_pvr: float,
                 observed_pvr: Optional[float], radius: Optional[float], successes: int, attempts: int):
        self.medoid = medoid
        self.seed = seed
        self.members = members
        self.maximal_pvr = maximal_pvr
        self.observed_pvr = observed_pvr
        self.radius = radius
        self.successes = successes
        self.attempts = attempts

    @property
    def kind_str(self) -> str:
        if self.observed_pvr is not None:
            return "normal"
        return "loner" if self.radius is None else "fallback"


class ScanStats:
    """What one medoid scan returns to the host (all exact)."""

    __slots__ = ("density", "n_within", "n_lt", "hist_fx", "list_ref")

    def __init__(self, density_fx: int, n_within: int, n_lt: int, hist_fx: _np.ndarray):
        # the python float the reference gets from `.sum().item()` on a float32 tensor (cluster.py:629)
        self.density = float(_np.float32(density_fx / _lib.DENSITY_SCALE))
        self.n_within = n_within
        self.n_lt = n_lt
        self.hist_fx = hist_fx
        # (scan sequence number, medoid slot) of the device-side list of rows within the medoid radius,
        # or None when the backend keeps no such list (then the host asks for a select pass)
        self.list_ref = None

    @classmethod
    def batch(cls, raw: _np.ndarray, seq=None) -> list:
        """ScanStats for every row of a raw accumulator block [k, 63] (vectorised conversions)."""
        dens = (raw[:, 0] / _lib.DENSITY_SCALE).astype(_np.float32).astype(_np.float64).tolist()
        n_within = raw[:, _NBINS + 1].tolist()
        n_lt = raw[:, _NBINS + 2].tolist()
        out = []
        for j in range(len(raw)):
            st = cls.__new__(cls)
            st.density = dens[j]
            st.n_within = n_within[j]
            st.n_lt = n_lt[j]
            st.hist_fx = raw[j, 1:_NBINS + 1]
            st.list_ref = None if seq is None else (seq, j)
            out.append(st)
        return out


class HipScanBackend:
    """Single-GPU backend: one ``vh_clu`` handle holding the whole matrix."""

    def __init__(self, matrix: _np.ndarray, lengths_f32: _np.ndarray, normalized: bool,
                 normalized_out: Optional[_np.ndarray]):
        self.lib = _lib.load()
        _lib.require_gpu()
        handle = ctypes.c_void_p()
        _lib.sync_env_options()   # VAMBHIP_* variables -> library options (the .so reads no environment)
        _lib.check(self.lib.vh_clu_create(_lib.ptr(matrix), _lib.ptr(lengths_f32), matrix.shape[0],
                                          matrix.shape[1], int(bool(normalized)), _lib.ptr(normalized_out),
                                          ctypes.byref(handle)))
        self.h = handle
        self.L = matrix.shape[1]
        self.n_rows = matrix.shape[0]
        self._res = (_lib.ScanResult * _MAX_MEDOIDS_PER_PASS)()
        k = ctypes.c_int(0)
        _lib.check(self.lib.vh_clu_max_medoids(self.h, ctypes.byref(k)))
        self.max_medoids = k.value          # < 32 for very wide latent spaces (query vectors are staged in LDS)
        self._sel = _np.empty(max(1, self.n_rows), _np.int64)
        # accounting for bench.py / DESIGN.md roofline: rows streamed by scan and select passes
        self.scan_passes = 0
        self.scan_medoids = 0
        self.rows_streamed = 0          # resident rows per pass (what the kernels read)
        self.live_rows_streamed = 0     # live rows per pass (algorithmic bytes)
        self.kernel_ms = 0.0
        self.timing = False

    def _n_live(self) -> int:
        n_live = ctypes.c_int64()
        _lib.check(self.lib.vh_clu_rows(self.h, None, ctypes.byref(n_live)))
        return n_live.value

    def close(self):
        if self.h is not None and self.h.value:
            gen = getattr(self, "gen_handle", None)   # the native state machine borrows this handle
            if gen is not None:
                self.lib.vh_gen_destroy(gen)
                self.gen_handle = None
            self.lib.vh_clu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on: bool):
        self.timing = bool(on)
        _lib.check(self.lib.vh_clu_set_timing(self.h, int(on)))

    def _collect_ms(self):
        if self.timing:
            ms = ctypes.c_float(0)
            _lib.check(self.lib.vh_clu_last_kernel_ms(self.h, ctypes.byref(ms)))
            self.kernel_ms += ms.value

    def scan_raw(self, rows, queries=None) -> _np.ndarray:
        """One pass for <= 32 medoids.  rows[j] = physical row whose distance is forced to 0 (or -1 when
        the medoid lives in another shard, then ``queries`` [k, L] must be given).  Returns the raw
        int64 accumulators [k, 63] = (density_fx, hist_fx[60], n_within, n_lt)."""
        rows = _np.ascontiguousarray(rows, dtype=_np.int64)
        k = len(rows)
        q = None if queries is None else _np.ascontiguousarray(queries, dtype=_np.float32)
        _lib.check(self.lib.vh_clu_scan(self.h, k, _lib.ptr(rows), _lib.ptr(q), self._res))
        self.scan_passes += 1
        self.scan_medoids += k
        self.rows_streamed += self.n_rows
        self.live_rows_streamed += self._n_live()
        self._collect_ms()
        return _np.frombuffer(self._res, dtype=_np.int64, count=k * (_NBINS + 3)).reshape(k, _NBINS + 3).copy()

    def scan(self, medoids):
        """List of physical rows -> list of ScanStats (one pass per <= 32 medoids)."""
        out = []
        seq = ctypes.c_int64(0)
        for lo in range(0, len(medoids), self.max_medoids):
            _lib.check(self.lib.vh_clu_scan_seq(self.h, ctypes.byref(seq)))
            raw = self.scan_raw(medoids[lo:lo + self.max_medoids])
            out.extend(ScanStats.batch(raw, seq.value))
        return out

    def scan_list(self, list_ref) -> Optional[_np.ndarray]:
        """Ascending rows within the medoid radius recorded by the scan `list_ref` came from, or None when
        that scan has left the device ring / the list overflowed (the caller then runs a select pass)."""
        n = ctypes.c_int64(0)
        _lib.check(self.lib.vh_clu_scan_list(self.h, int(list_ref[0]), int(list_ref[1]), _lib.ptr(self._sel),
                                             len(self._sel), ctypes.byref(n)))
        if n.value < 0:
            return None
        return self._sel[: n.value].copy()

    def get_rows(self, rows) -> _np.ndarray:
        rows = _np.ascontiguousarray(rows, dtype=_np.int64)
        out = _np.empty((len(rows), self.L), _np.float32)
        if len(rows):
            _lib.check(self.lib.vh_clu_get_rows(self.h, _lib.ptr(rows), len(rows), _lib.ptr(out)))
        return out

    def select_query(self, row: int, query, threshold: float, remove: bool) -> _np.ndarray:
        """select() against an explicit query vector (row = -1 when the medoid is not in this shard)."""
        n = ctypes.c_int64(0)
        thr = float(_np.float32(threshold))
        q = None if query is None else _np.ascontiguousarray(query, dtype=_np.float32)
        _lib.check(self.lib.vh_clu_select(self.h, int(row), _lib.ptr(q), thr, int(remove), _lib.ptr(self._sel),
                                          len(self._sel), ctypes.byref(n)))
        self.rows_streamed += self.n_rows
        self.live_rows_streamed += self._n_live() + (n.value if remove else 0)
        self.scan_passes += 1
        self._collect_ms()
        return self._sel[: n.value].copy()

    def select(self, medoid: int, threshold: float, remove: bool) -> _np.ndarray:
        # torch compares in float32 (cluster.py:640-650): select_query casts the threshold
        return self.select_query(int(medoid), None, threshold, remove)

    def remove(self, rows: _np.ndarray):
        rows = _np.ascontiguousarray(rows, dtype=_np.int64)
        _lib.check(self.lib.vh_clu_remove(self.h, _lib.ptr(rows), len(rows)))

    def pack(self) -> int:
        n = ctypes.c_int64(0)
        _lib.check(self.lib.vh_clu_pack(self.h, ctypes.byref(n)))
        self.n_rows = n.value
        return n.value

    def matrix(self) -> _np.ndarray:
        out = _np.empty((self.n_rows, self.L), _np.float32)
        _lib.check(self.lib.vh_clu_get_rows(self.h, None, 0, _lib.ptr(out)))
        return out


class _MatrixView:
    """``ClusterGenerator.matrix`` -- supports ``.numpy()`` and ``len()`` like the reference's tensor."""

    def __init__(self, backend, sync=None):
        self._backend = backend
        self._sync = sync          # refreshes backend.n_rows when the native state machine has packed the matrix

    def numpy(self) -> _np.ndarray:
        if self._sync is not None:
            self._sync()
        return self._backend.matrix()

    def __len__(self):
        if self._sync is not None:
            self._sync()
        return self._backend.n_rows


def smooth_histogram(histogram: _np.ndarray) -> _np.ndarray:
    """31-tap smoothing of the 60-bin histogram, float32 multiply then float32 add in ascending bin
    order (cluster.py:495-500); returns the 60 densities of cluster.py:500."""
    # densities[k] = sum_i pdf[k - i] * hist[i], products and running sum in float32, i ascending.  Walking
    # the taps from the last to the first adds, for every k, the terms in ascending i -- the order of the
    # reference's `densities[i:i+31] += pdf * histogram[i]` loop -- with 31 vector adds instead of 60.
    pdf_len = len(_NORMALPDF)
    n = len(histogram)
    histogram = _np.asarray(histogram, dtype=_np.float32)
    densities = _np.zeros(n + pdf_len - 1, dtype=_np.float32)
    for t in range(pdf_len - 1, -1, -1):
        densities[t:t + n] += _NORMALPDF[t] * histogram
    return densities[15:-15]


def threshold_from_densities(densities: _np.ndarray, peak_valley_ratio: float):
    """Peak / valley walk of cluster.py:483-543.  NoThreshold() or (threshold, observed_pvr)."""
    peak_density = 0.0
    peak_over = False
    minimum_x = 0.0
    threshold = None
    delta_x = _XMAX / len(densities)
    x = 0
    density_at_minimum = 0.0
    for value in densities.tolist():  # float32 -> python float, as `.item()` does
        density = value
        if not peak_over and density > peak_density:
            if x > 0.1:  # first peak must not lie beyond 0.1
                return NoThreshold()
            peak_density = density
        if not peak_over and density < 0.6 * peak_density:
            peak_over = True
            density_at_minimum = density
        if peak_over and density > 1.5 * density_at_minimum:
            break
        if peak_over and density < density_at_minimum:
            minimum_x, density_at_minimum = x, density
            if density < peak_valley_ratio * peak_density:
                threshold = minimum_x
        x += delta_x
    if threshold is None or threshold > 0.2 + peak_valley_ratio:
        return NoThreshold()
    return (threshold, density_at_minimum / peak_density)


class ClusterGenerator:
    """Iterative medoid cluster generator running its matrix passes on an MI355X.

    Inputs (identical to the reference, cluster.py:122-135, 234-245):
        matrix: (obs x features) numpy float32
        lengths: contig lengths
        maxsteps, windowsize, minsuccesses: search parameters
        destroy: normalise ``matrix`` in place instead of copying
        normalized: matrix is already preprocessed
        cuda: accepted for signature compatibility; the matrix passes always run on the GPU
        rng_seed: seed of the candidate sampler
    """

    # compact the resident matrix when fewer than this fraction of rows are live.  Measured at C2 (profiles/r04e_*, cached
    # statistics survive a pack since round 4): 0.5 -> 14.6 s, 0.75 -> 14.2 s, 0.85 -> 13.6 s, 0.92 -> 13.3 s per sweep
    PACK_FRACTION = 0.9
    PACK_MIN_ROWS = 8192

    def __repr__(self) -> str:
        return f"ClusterGenerator({len(self.matrix)} points, {self.n_emitted_clusters} clusters)"

    def __str__(self) -> str:
        return (f"ClusterGenerator({len(self.matrix)} points, {self.n_emitted_clusters} clusters)\n"
                f"  CUDA:         {self.cuda}\n  maxsteps:     {self.maxsteps}\n"
                f"  minsuccesses: {self.minsuccesses}\n  pvr:          {self.peak_valley_ratio}\n"
                f"  successes:    {self.successes}/{self._n_attempts()}\n")

    def _n_attempts(self) -> int:
        return self._native_attempts if self._gen is not None else len(self.attempts)

    @staticmethod
    def _check_params(matrix, lengths, maxsteps, windowsize, minsuccesses) -> None:
        if matrix.dtype != _np.float32:
            raise ValueError("Matrix must be of dtype float32")
        if maxsteps < 1:
            raise ValueError(f"maxsteps must be a positive integer, not {maxsteps}")
        if windowsize < 1:
            raise ValueError(f"windowsize must be at least 1, not {windowsize}")
        if minsuccesses < 1 or minsuccesses > windowsize:
            raise ValueError(f"minsuccesses must be between 1 and windowsize, not {minsuccesses}")
        if len(matrix) < 1:
            raise ValueError("Matrix must have at least 1 observation.")
        if len(lengths) != len(matrix):
            raise ValueError("N sequences in lengths and matrix do not match")

    def __init__(self, matrix: _np.ndarray, lengths: _np.ndarray, maxsteps: int = 25, windowsize: int = 300,
                 minsuccesses: int = 15, destroy: bool = False, normalized: bool = False, cuda: bool = False,
                 rng_seed: int = 0, _backend_factory=None):
        self._check_params(matrix, lengths, maxsteps, windowsize, minsuccesses)
        if matrix.ndim != 2:
            raise ValueError("Matrix must be 2-dimensional")
        lengths = _np.asarray(lengths)
        lengths_f32 = _np.ascontiguousarray(lengths, dtype=_np.float32)  # torch.Tensor(lengths), cluster.py:277

        inplace_target = None
        upload = matrix
        if not matrix.flags.c_contiguous:
            upload = _np.ascontiguousarray(matrix)
            if destroy:
                inplace_target = matrix
        # destroy=True: the caller's array receives the normalised rows (cluster.py:253-258)
        normalized_out = upload if (destroy and not normalized) else None

        factory = HipScanBackend if _backend_factory is None else _backend_factory
        self._backend = factory(upload, lengths_f32, normalized, normalized_out)
        if inplace_target is not None and normalized_out is not None:
            inplace_target[...] = normalized_out

        self._setup(lengths, maxsteps, windowsize, minsuccesses, rng_seed, native=_backend_factory is None)

    @classmethod
    def from_backend(cls, backend, lengths: _np.ndarray, maxsteps: int = 25, windowsize: int = 300,
                     minsuccesses: int = 15, rng_seed: int = 0):
        """Build the host state machine on top of an existing scan backend (the row-sharded multi-GPU
        backend of ``vamb_amd.parallel``).  ``lengths`` are the GLOBAL contig lengths."""
        lengths = _np.asarray(lengths)
        if maxsteps < 1:
            raise ValueError(f"maxsteps must be a positive integer, not {maxsteps}")
        if windowsize < 1:
            raise ValueError(f"windowsize must be at least 1, not {windowsize}")
        if minsuccesses < 1 or minsuccesses > windowsize:
            raise ValueError(f"minsuccesses must be between 1 and windowsize, not {minsuccesses}")
        if backend.n_rows < 1:
            raise ValueError("Matrix must have at least 1 observation.")
        if len(lengths) != backend.n_rows:
            raise ValueError("N sequences in lengths and matrix do not match")
        self = cls.__new__(cls)
        self._backend = backend
        self._setup(lengths, maxsteps, windowsize, minsuccesses, rng_seed, native=True)
        return self

    def _setup(self, lengths, maxsteps, windowsize, minsuccesses, rng_seed, native=False):
        self._gen = None
        if _os.environ.get("VAMBHIP_PACK_FRACTION"):     # A/B runs of the packing policy (results do not depend on it)
            self.PACK_FRACTION = float(_os.environ["VAMBHIP_PACK_FRACTION"])
        self.maxsteps: int = maxsteps
        self.minsuccesses: int = minsuccesses
        self.cuda: bool = True
        self.rng = _random.Random(rng_seed)
        self.matrix = _MatrixView(self._backend, self._sync_native_counters)
        n = self._backend.n_rows
        self.indices = _np.arange(n)                       # original row of every resident row
        self._kept = _np.ones(n, dtype=bool)               # host mirror of the device live mask
        self._alive = _np.ones(n, dtype=bool)              # the same, indexed by ORIGINAL contig index
        self.order = _np.argsort(lengths)[::-1].copy()     # same call as cluster.py:275
        self.order_index = 0
        self.n_emitted_clusters = 0
        self.n_remaining_points = n
        self.peak_valley_ratio = 0.1
        self.attempts: _deque = _deque(maxlen=windowsize)
        self.successes = 0
        self._stats_cache: dict[int, ScanStats] = {}
        self._within_cache: dict[int, _np.ndarray] = {}
        # Single-GPU handles run the state machine below in native code (vh_gen_*, same scans, same
        # random.Random stream); the Python methods remain the implementation for every other backend
        # (row-sharded multi-GPU, the CPU oracle backend of the tests) and the specification of both.
        self._gen = None
        self._members_buf = None
        self._native_attempts = 0
        self._native_queue = _deque()     # clusters the native state machine has produced ahead of the iteration
        self._sharded_native = False
        seed_ok = isinstance(rng_seed, int) and abs(rng_seed) < 2 ** 64 and not _os.environ.get("VAMBHIP_PY_GENERATOR")
        local = getattr(self._backend, "local", None)      # vamb_amd.parallel.ShardedScanBackend: this rank's shard
        if native and seed_ok and isinstance(self._backend, HipScanBackend):
            handle = ctypes.c_void_p()
            order = _np.ascontiguousarray(self.order, dtype=_np.int64)
            _lib.sync_env_options()   # VAMBHIP_* variables -> library options (the .so reads no environment)
            _lib.check(self._backend.lib.vh_gen_create(self._backend.h, _lib.ptr(order), n, int(maxsteps),
                                                       int(windowsize), int(minsuccesses), abs(rng_seed),
                                                       float(self.PACK_FRACTION), int(self.PACK_MIN_ROWS),
                                                       ctypes.byref(handle)))
            self._gen = handle
            self._backend.gen_handle = handle      # destroyed by the backend, before the handle it borrows
            self._members_buf = _np.empty(n, _np.int64)
        elif (native and seed_ok and isinstance(local, HipScanBackend) and getattr(self._backend, "device_plane", False)):
            # row-sharded matrix, HIP shards, a device data plane: the native state machine over all shards (collective:
            # every rank creates it and iterates in lock step); n = GLOBAL rows
            handle = ctypes.c_void_p()
            order = _np.ascontiguousarray(self.order, dtype=_np.int64)
            _lib.sync_env_options()
            _lib.check(local.lib.vh_gen_create_sharded(local.h, _lib.ptr(order), n, int(maxsteps), int(windowsize),
                                                       int(minsuccesses), abs(rng_seed), float(self.PACK_FRACTION),
                                                       int(self.PACK_MIN_ROWS), ctypes.byref(handle)))
            self._gen = handle
            local.gen_handle = handle
            self._members_buf = _np.empty(n, _np.int64)
            self._sharded_native = True

    _NATIVE_BATCH = 64     # clusters fetched per call of the native state machine (vh_gen_next_batch)

    # vh_cluster_info as a numpy record: a batch is unpacked column by column (13 ctypes field reads per cluster cost the sweep
    # ~0.2 s of host time at 2 M contigs / 210 k clusters, with the GPU idle)
    _INFO_DTYPE = _np.dtype([(name, {ctypes.c_int64: "<i8", ctypes.c_int32: "<i4", ctypes.c_double: "<f8"}[ctype])
                             for name, ctype in _lib.ClusterInfo._fields_])
    assert _INFO_DTYPE.itemsize == ctypes.sizeof(_lib.ClusterInfo)

    def _next_native(self) -> Cluster:
        if not self._native_queue:
            lib = self._backend.lib
            if getattr(self, "_infos", None) is None:
                self._infos = _np.zeros(self._NATIVE_BATCH, self._INFO_DTYPE)
            n = ctypes.c_int(0)
            _lib.check(lib.vh_gen_next_batch(self._gen, self._NATIVE_BATCH,
                                             self._infos.ctypes.data_as(ctypes.POINTER(_lib.ClusterInfo)),
                                             _lib.ptr(self._members_buf), len(self._members_buf), ctypes.byref(n)))
            if n.value == 0:
                self._sync_native_counters()
                raise StopIteration
            rec = self._infos[:n.value]
            sizes = rec["n_members"]
            ends = _np.cumsum(sizes)
            members = _np.split(self._members_buf[:int(ends[-1])].copy(), ends[:-1])   # one copy per batch, a view per cluster
            col = {name: rec[name].tolist() for name in rec.dtype.names}
            self._native_queue.extend(zip(col["medoid"], col["seed"], members, col["kind"], col["maximal_pvr"], col["observed_pvr"],
                                          col["radius"], col["successes"], col["attempts"], col["pvr_after"], col["successes_after"],
                                          col["attempts_after"], col["order_index_after"]))
            self._counters_stale = True
        (medoid, seed, members, kind, maximal_pvr, observed_pvr, radius, successes, attempts, pvr_after, successes_after,
         attempts_after, order_index_after) = self._native_queue.popleft()
        self.n_emitted_clusters += 1
        self.n_remaining_points -= len(members)
        # the attributes the reference exposes (repr, callers inspecting the search state) AFTER update_successes
        self.peak_valley_ratio = pvr_after
        self.successes = successes_after
        self._native_attempts = attempts_after
        self.order_index = order_index_after
        return Cluster(medoid, seed, members, maximal_pvr, observed_pvr if kind == 0 else None,
                       None if kind == 1 else radius, successes, attempts)

    def _sync_native_counters(self):
        """Mirror the native counters on the Python objects (bench accounting, repr, len(matrix))."""
        if self._gen is None or not getattr(self, "_counters_stale", False):
            return
        b, lib = self._backend, self._backend.lib
        passes, medoids, rows = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        emitted, remaining, ms = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.vh_gen_counters(self._gen, ctypes.byref(passes), ctypes.byref(medoids), ctypes.byref(rows),
                                       ctypes.byref(ms), ctypes.byref(emitted), ctypes.byref(remaining)))
        b.scan_passes, b.scan_medoids, b.rows_streamed, b.kernel_ms = passes.value, medoids.value, rows.value, ms.value
        live = ctypes.c_int64()
        _lib.check(lib.vh_gen_live_rows(self._gen, ctypes.byref(live)))
        b.live_rows_streamed = live.value
        if not self._sharded_native:   # (a sharded backend's n_rows is the GLOBAL row count: left to its own bookkeeping)
            n_rows, n_live = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.vh_clu_rows(b.h, ctypes.byref(n_rows), ctypes.byref(n_live)))
            b.n_rows = n_rows.value
        self._counters_stale = False

    def __iter__(self):
        return self

    # cluster.py:298-316
    def __next__(self) -> Cluster:
        if self._gen is not None:
            return self._next_native()
        if self.n_remaining_points == 0:
            raise StopIteration
        assert self.n_remaining_points > 0
        cluster, points = self._find_cluster()
        self._stats_cache.clear()
        self._within_cache.clear()
        self.n_emitted_clusters += 1
        self.n_remaining_points -= len(points)
        self._kept[points] = False
        self._alive[self.indices[points]] = False
        n_rows = len(self._kept)
        if (self.n_remaining_points > 0 and n_rows >= self.PACK_MIN_ROWS
                and self.n_remaining_points < self.PACK_FRACTION * n_rows):
            self.pack()
        return cluster

    # cluster.py:318-335
    def pack(self):
        "Remove all used points from the resident matrix and indices."
        new_n = self._backend.pack()
        self.indices = self.indices[self._kept]
        assert new_n == len(self.indices)
        self._kept = _np.ones(new_n, dtype=bool)
        self._stats_cache.clear()
        self._within_cache.clear()

    def pack_order(self):
        self.order = self.order[self.order > -1]
        assert len(self.order) > 0

    # cluster.py:342-384
    def get_next_seed(self) -> int:
        """Next live contig in descending-length order.  Same walk as the reference (dead entries are
        overwritten with -1 as they are passed, the order array is compacted on every wrap-around once a
        cluster has been emitted), but dead runs are skipped a vectorised chunk at a time."""
        n_order = len(self.order)
        i = self.order_index % n_order
        chunk = 64
        while True:
            if i == 0 and self.n_emitted_clusters > 0:
                self.pack_order()
                n_order = len(self.order)
            hi = min(i + chunk, n_order)
            seg = self.order[i:hi]
            live = seg > -1
            live[live] = self._alive[seg[live]]
            if live.any():
                pos = int(live.argmax())
                seg[:pos] = -1
                self.order_index = i + pos + 1
                row = int(_np.searchsorted(self.indices, seg[pos]))
                assert self.indices[row] == seg[pos] and self._kept[row]
                return row
            seg[:] = -1
            i = hi % n_order
            chunk = min(chunk * 4, 1 << 16)

    # cluster.py:386-413
    def update_successes(self, success: bool):
        if len(self.attempts) == self.attempts.maxlen:
            self.successes -= self.attempts.popleft()
        self.successes += success
        self.attempts.append(success)
        if len(self.attempts) == self.attempts.maxlen and self.successes < self.minsuccesses:
            self.peak_valley_ratio += 0.1
            self.attempts.clear()
            self.successes = 0
            self.order_index = 0

    # ---- device-backed pieces of sample_medoid (cluster.py:606-637) ----------------------------
    def _ensure_stats(self, medoids):
        missing = [m for m in dict.fromkeys(medoids) if m not in self._stats_cache]
        if missing:
            for m, st in zip(missing, self._backend.scan(missing)):
                self._stats_cache[m] = st

    def _within(self, medoid: int) -> _np.ndarray:
        hit = self._within_cache.get(medoid)
        if hit is None:
            st = self._stats_cache.get(medoid)
            fetch = getattr(self._backend, "scan_list", None)
            if st is not None and st.list_ref is not None and fetch is not None:
                hit = fetch(st.list_ref)
                if hit is not None:
                    assert len(hit) == st.n_within
            if hit is None:
                hit = self._backend.select(medoid, _MEDOID_RADIUS, remove=False)
            self._within_cache[medoid] = hit
        return hit

    def sample_medoid(self, medoid: int):
        """(rows within 0.05 of `medoid`, its ScanStats, local density) -- cluster.py:606-637."""
        self._ensure_stats([medoid])
        st = self._stats_cache[medoid]
        return self._within(medoid), st, st.density

    # cluster.py:415-450
    def wander_medoid(self, seed: int):
        medoid = seed
        tried = {medoid}
        self._ensure_stats([seed])
        stats = self._stats_cache[seed]
        local_density = stats.density
        candidates = [i for i in self._within(seed).tolist() if i not in tried]
        candidates = self.rng.sample(candidates, k=min(len(candidates), self.maxsteps))
        i = 0
        while i < len(candidates):
            # look ahead: every not-yet-scanned candidate of this round shares one matrix pass
            self._ensure_stats(candidates[i:])
            sampled = candidates[i]
            tried.add(sampled)
            sampled_stats = self._stats_cache[sampled]
            if sampled_stats.density > local_density:
                medoid, stats, local_density = sampled, sampled_stats, sampled_stats.density
                candidates = [j for j in self._within(sampled).tolist() if j not in tried]
                candidates = self.rng.sample(candidates, k=min(len(candidates), self.maxsteps))
                i = 0
            else:
                i += 1
        return medoid, stats

    # cluster.py:452-543
    def find_threshold(self, stats: ScanStats):
        if stats.n_lt == 1:
            return Loner()
        histogram = (stats.hist_fx.astype(_np.float64) / _lib.HIST_SCALE).astype(_np.float32)
        return threshold_from_densities(smooth_histogram(histogram), self.peak_valley_ratio)

    def _logical_index(self, row: int) -> int:
        """Index the row would have in the reference's packed matrix (reported as Cluster.seed)."""
        return int(_np.count_nonzero(self._kept[:row]))

    # cluster.py:545-604
    def _find_cluster(self):
        while True:
            seed = self.get_next_seed()
            medoid, stats = self.wander_medoid(seed)
            threshold = self.find_threshold(stats)
            original = int(self.indices[medoid])
            if isinstance(threshold, Loner):
                cluster = Cluster(original, self._logical_index(seed), _np.array([original]),
                                  self.peak_valley_ratio, None, None, self.successes, len(self.attempts))
                points = _np.array([medoid], dtype=_np.int64)
                self._backend.remove(points)
                return cluster, points
            if isinstance(threshold, NoThreshold):
                if self.peak_valley_ratio > 0.55:
                    seed_logical = self._logical_index(seed)
                    points = self._backend.select(medoid, _DEFAULT_RADIUS, remove=True)
                    cluster = Cluster(original, seed_logical, self.indices[points], self.peak_valley_ratio,
                                      None, _DEFAULT_RADIUS, self.successes, len(self.attempts))
                    return cluster, points
                self.update_successes(False)
                continue
            radius, observed_pvr = threshold
            seed_logical = self._logical_index(seed)
            points = self._backend.select(medoid, radius, remove=True)
            cluster = Cluster(original, seed_logical, self.indices[points], self.peak_valley_ratio,
                              observed_pvr, radius, self.successes, len(self.attempts))
            if self.peak_valley_ratio < 0.55:
                self.update_successes(True)
            return cluster, points
