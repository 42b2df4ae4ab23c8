"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ``vamb.cluster.ClusterGenerator`` (cuda=False path).

Part of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this.  The product (``vamb_amd/``) never does.

Control flow follows the reference's CPU path line by line (citations below, all into
``/root/reference/vamb/cluster.py``); the floating-point arithmetic of the distance scan is the
defined-order version in ``oracle/cluster_scan.c`` (see that file's header for the contract).  The
restatement is pinned against the reference itself: ``tests/golden/make_golden.py`` runs the real
``ClusterGenerator`` (torch CPU/MKL) on seeded fixtures and commits the cluster streams;
``tests/test_oracle_cluster.py`` requires this oracle to reproduce them exactly.
"""
from __future__ import annotations

import ctypes
import os
import random
from collections import OrderedDict, deque

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NBINS = 60
DENSITY_SCALE = 65536.0
HIST_SCALE = 256.0
MEDOID_RADIUS = 0.05      # cluster.py:15
DEFAULT_RADIUS = 0.06     # cluster.py:13
DELTA_X = 0.005           # cluster.py:17
XMAX = 0.3                # cluster.py:18
MAX_CACHED = 64           # cluster.py:22

# cluster.py:39-73: N(0, 0.01) pdf sampled every 0.005 on [-0.075, 0.075]; the reference builds it as
# float32 tensor times python 0.005, i.e. float32 * float32.
_PDF_TABLE = np.array(
    [2.43432053e-11, 9.13472041e-10, 2.66955661e-08, 6.07588285e-07, 1.07697600e-05,
     1.48671951e-04, 1.59837411e-03, 1.33830226e-02, 8.72682695e-02, 4.43184841e-01,
     1.75283005e00, 5.39909665e00, 1.29517596e01, 2.41970725e01, 3.52065327e01,
     3.98942280e01, 3.52065327e01, 2.41970725e01, 1.29517596e01, 5.39909665e00,
     1.75283005e00, 4.43184841e-01, 8.72682695e-02, 1.33830226e-02, 1.59837411e-03,
     1.48671951e-04, 1.07697600e-05, 6.07588285e-07, 2.66955661e-08, 9.13472041e-10,
     2.43432053e-11], dtype=np.float32)
NORMALPDF = (np.float32(DELTA_X) * _PDF_TABLE).astype(np.float32)


def lib():
    """Load oracle/liboracle.so (built by ``__graft_entry__.build()`` / ``oracle/Makefile``)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` or __graft_entry__.build()")
        L = ctypes.CDLL(path)
        i64, vp, f32 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
        L.vo_normalize.argtypes = [vp, i64, ctypes.c_int]
        L.vo_set_order.argtypes = [ctypes.c_int]
        L.vo_torch_sum.argtypes = [vp, i64]
        L.vo_torch_sum.restype = f32
        L.vo_reference_sums.argtypes = [vp, vp, vp, i64, vp, vp, vp]
        L.vo_get_order.restype = ctypes.c_int
        L.vo_distances.argtypes = [vp, i64, ctypes.c_int, i64, vp]
        L.vo_scan.argtypes = [vp, vp, vp, i64, ctypes.c_int, i64, vp, vp, vp, vp, vp, vp, i64]
        L.vo_select.argtypes = [vp, vp, i64, ctypes.c_int, i64, f32, vp, i64]
        L.vo_select.restype = i64
        L.vo_scan_q.argtypes = [vp, vp, vp, i64, ctypes.c_int, i64, vp, vp, vp, vp, vp, vp, vp, i64]
        L.vo_select_q.argtypes = [vp, vp, i64, ctypes.c_int, i64, vp, f32, vp, i64]
        L.vo_select_q.restype = i64
        L.vo_compact_rows.argtypes = [vp, vp, i64, ctypes.c_int]
        L.vo_compact_rows.restype = i64
        L.vo_bin.argtypes = [f32]
        L.vo_bin.restype = ctypes.c_int
        L.vo_edges.argtypes = [vp]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


DEFAULT_ORDER = 2


def set_order(order: int) -> None:
    """Evaluation order of the two float32 reductions behind every decision (cluster_scan.c header):
    2 (default; == libvambhip's default, scan.reference_order = 2): `matmul` and `norm` as the reference's own torch / oneMKL
      AVX-512 CPU build evaluates them -- distances and normalisation equal the real reference's bit for bit -- with the exact
      integer density / histogram sums;
    1: the same, and the density / histogram sums in float32 in the reference's own order (torch.sum / one-thread
      torch.histogram): every reported field of the stream, observed_pvr included, is the reference's;
    0: the ascending fmaf chain (libvambhip: scan.reference_order = 0)."""
    lib().vo_set_order(int(order))


def get_order() -> int:
    return int(lib().vo_get_order())


def normalize(matrix: np.ndarray) -> np.ndarray:
    """cluster.py:653-669, in place on a C-contiguous float32 [N, L] array."""
    assert matrix.dtype == np.float32 and matrix.flags.c_contiguous
    lib().vo_normalize(_p(matrix), matrix.shape[0], matrix.shape[1])
    return matrix


def scan(matrix, lengths_f32, kept, medoid, want_dist=True):
    """One ``sample_medoid`` (cluster.py:606-637) plus the histogram inputs of ``find_threshold``
    (cluster.py:457-481) for row ``medoid``.  Returns a dict of raw (integer) accumulators."""
    n, L = matrix.shape
    dist = np.empty(n, np.float32) if want_dist else None
    hist = np.zeros(NBINS, np.int64)
    dens = np.zeros(1, np.int64)
    nw = np.zeros(1, np.int64)
    nlt = np.zeros(1, np.int64)
    within = np.empty(n, np.int64)
    lib().vo_scan(_p(matrix), _p(lengths_f32), _p(kept), n, L, int(medoid), _p(dist), _p(hist),
                  _p(dens), _p(nw), _p(nlt), _p(within), n)
    return dict(dist=dist, hist_fx=hist, density_fx=int(dens[0]), n_within=int(nw[0]),
                n_lt=int(nlt[0]), within=within[: int(nw[0])].copy())


def scan_query(matrix, lengths_f32, kept, medoid, query):
    """scan() against an explicit query vector; medoid = -1 when the medoid row is in another shard."""
    n, L = matrix.shape
    hist = np.zeros(NBINS, np.int64)
    dens = np.zeros(1, np.int64)
    nw = np.zeros(1, np.int64)
    nlt = np.zeros(1, np.int64)
    q = np.ascontiguousarray(query, dtype=np.float32)
    lib().vo_scan_q(_p(matrix), _p(lengths_f32), _p(kept), n, L, int(medoid), _p(q), None, _p(hist), _p(dens),
                    _p(nw), _p(nlt), None, 0)
    return dict(hist_fx=hist, density_fx=int(dens[0]), n_within=int(nw[0]), n_lt=int(nlt[0]))


def select_query(matrix, kept, medoid, query, threshold):
    n, L = matrix.shape
    out = np.empty(max(n, 1), np.int64)
    q = np.ascontiguousarray(query, dtype=np.float32)
    cnt = lib().vo_select_q(_p(matrix), _p(kept), n, L, int(medoid), _p(q), float(np.float32(threshold)), _p(out), n)
    return out[:cnt].copy()


def select(matrix, kept, medoid, threshold):
    """cluster.py:640-650 with a float32 compare (torch casts the python threshold to the tensor dtype)."""
    n, L = matrix.shape
    out = np.empty(n, np.int64)
    cnt = lib().vo_select(_p(matrix), _p(kept), n, L, int(medoid), float(np.float32(threshold)), _p(out), n)
    return out[:cnt].copy()


def torch_sum(x: np.ndarray) -> float:
    """``torch.sum`` of a contiguous float32 vector in ATen's evaluation order (cluster_scan.c vo_torch_sum)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().vo_torch_sum(_p(x), len(x)))


def reference_sums(dist, lengths_f32, kept):
    """(local_density, histogram) as the reference itself computes them from these distances: torch.sum's order for the density,
    torch.histogram's one-thread order for the bins (cluster_scan.c vo_reference_sums)."""
    n = len(dist)
    scratch = np.empty(max(n, 1), np.float32)
    dens = np.zeros(1, np.float32)
    hist = np.zeros(NBINS, np.float32)
    lib().vo_reference_sums(_p(dist), _p(lengths_f32), _p(kept), n, _p(scratch), _p(dens), _p(hist))
    return float(dens[0]), hist


def density_value(density_fx: int) -> float:
    """The python float the reference would get from ``.sum().item()`` of a float32 tensor."""
    return float(np.float32(density_fx / DENSITY_SCALE))


def histogram_value(hist_fx: np.ndarray) -> np.ndarray:
    return (hist_fx.astype(np.float64) / HIST_SCALE).astype(np.float32)


def smooth(histogram_f32: np.ndarray) -> np.ndarray:
    """cluster.py:495-500: 31-tap smoothing, float32 multiply then float32 add, bins ascending."""
    pdf_len = len(NORMALPDF)
    dens = np.zeros(len(histogram_f32) + pdf_len - 1, np.float32)
    for i in range(len(histogram_f32)):
        dens[i:i + pdf_len] += NORMALPDF * histogram_f32[i]
    return dens[15:-15]


def pick_threshold(densities_f32: np.ndarray, peak_valley_ratio: float):
    """cluster.py:483-543 peak/valley walk.  Returns None (NoThreshold) or (threshold, observed_pvr)."""
    peak_density = 0.0
    peak_over = False
    minimum_x = 0.0
    threshold = None
    delta_x = XMAX / len(densities_f32)
    x = 0
    density_at_minimum = 0.0
    for density_ in densities_f32:
        density = float(density_)
        if not peak_over and density > peak_density:
            if x > 0.1:
                return None
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
    if threshold is None:
        return None
    if threshold > 0.2 + peak_valley_ratio:
        return None
    return (threshold, density_at_minimum / peak_density)


class OracleCluster:
    __slots__ = ("medoid", "seed", "members", "maximal_pvr", "observed_pvr", "radius", "successes",
                 "attempts")

    def __init__(self, medoid, seed, members, maximal_pvr, observed_pvr, radius, successes, attempts):
        self.medoid = medoid
        self.seed = seed
        self.members = members
        self.maximal_pvr = maximal_pvr
        self.observed_pvr = observed_pvr
        self.radius = radius
        self.successes = successes
        self.attempts = attempts

    @property
    def kind_str(self):  # cluster.py:110-119
        if self.observed_pvr is not None:
            return "normal"
        return "loner" if self.radius is None else "fallback"


class OracleClusterGenerator:
    """Restatement of the reference iterator, CPU semantics (matrix packed after every cluster)."""

    def __init__(self, matrix, lengths, maxsteps=25, windowsize=300, minsuccesses=15, destroy=False,
                 normalized=False, rng_seed=0):
        # cluster.py:194-222
        if matrix.dtype != np.float32:
            raise ValueError("Matrix must be of dtype float32")
        if maxsteps < 1:
            raise ValueError("maxsteps must be a positive integer")
        if windowsize < 1:
            raise ValueError("windowsize must be at least 1")
        if minsuccesses < 1 or minsuccesses > windowsize:
            raise ValueError("minsuccesses must be between 1 and windowsize")
        if len(matrix) < 1:
            raise ValueError("Matrix must have at least 1 observation.")
        if len(lengths) != len(matrix):
            raise ValueError("N sequences in lengths and matrix do not match")
        if not destroy:
            matrix = matrix.copy()
        matrix = np.ascontiguousarray(matrix)
        if not normalized:
            normalize(matrix)
        self.matrix = matrix
        self.n = len(matrix)                      # live (packed) row count
        self.maxsteps = maxsteps
        self.minsuccesses = minsuccesses
        self.rng = random.Random(rng_seed)        # cluster.py:269
        self.indices = np.arange(self.n)          # cluster.py:274
        self.order = np.argsort(lengths)[::-1].copy()  # cluster.py:275
        self.order_index = 0
        self.lengths = np.asarray(lengths).astype(np.float32)  # torch.Tensor(lengths): cluster.py:277
        self.n_emitted_clusters = 0
        self.n_remaining_points = self.n
        self.peak_valley_ratio = 0.1
        self.attempts = deque(maxlen=windowsize)
        self.successes = 0
        self.cache = OrderedDict()
        self.n_scans = 0                          # bookkeeping for bench (bytes = sum of live rows)
        self.scan_rows = 0

    def __iter__(self):
        return self

    # cluster.py:298-316 + 318-335
    def __next__(self):
        if self.n_remaining_points == 0:
            raise StopIteration
        cluster, points = self._find_cluster()
        self.cache.clear()
        self.n_emitted_clusters += 1
        self.n_remaining_points -= len(points)
        kept = np.ones(self.n, np.uint8)
        kept[points] = 0
        newn = lib().vo_compact_rows(_p(self.matrix), _p(kept), self.n, self.matrix.shape[1])
        keepb = kept.astype(bool)
        self.indices = self.indices[keepb]
        self.lengths = np.ascontiguousarray(self.lengths[keepb])
        self.n = int(newn)
        return cluster

    def _live(self):
        return self.matrix[: self.n]

    # cluster.py:342-384 (CPU branches only)
    def _next_seed(self):
        n_order = len(self.order)
        i = self.order_index - 1
        while True:
            i = (i + 1) % n_order
            if i == 0 and self.n_emitted_clusters > 0:
                self.order = self.order[self.order > -1]
                assert len(self.order) > 0
                n_order = len(self.order)
            o = self.order[i]
            if o == -1:
                continue
            new_index = int(np.searchsorted(self.indices, o))
            if new_index >= len(self.indices) or self.indices[new_index] != o:
                self.order[i] = -1
                continue
            self.order_index = i + 1
            return new_index

    # cluster.py:386-413
    def _update_successes(self, success):
        if len(self.attempts) == self.attempts.maxlen:
            self.successes -= self.attempts.popleft()
        self.successes += success
        self.attempts.append(success)
        if len(self.attempts) == self.attempts.maxlen and self.successes < self.minsuccesses:
            self.peak_valley_ratio += 0.1
            self.attempts.clear()
            self.successes = 0
            self.order_index = 0

    # cluster.py:606-637
    def _sample(self, medoid):
        hit = self.cache.get(medoid)
        if hit is not None:
            return hit
        r = scan(self._live(), self.lengths, None, medoid)
        self.n_scans += 1
        self.scan_rows += self.n
        if get_order() == 1:
            # the reference's own float32 sums (torch.sum / one-thread torch.histogram order) instead of the exact ones: with
            # them every reported field of the stream, observed_pvr included, is the reference's bit for bit
            dens, hist = reference_sums(r["dist"], self.lengths, None)
            r["hist_ref"] = hist
            res = (r["within"], r, dens)
        else:
            res = (r["within"], r, density_value(r["density_fx"]))
        if len(self.cache) == MAX_CACHED:
            self.cache.popitem(last=False)
        self.cache[medoid] = res
        return res

    # cluster.py:415-450
    def _wander(self, seed):
        medoid = seed
        tried = {medoid}
        cluster, result, local_density = self._sample(seed)
        candidates = [i for i in cluster.tolist() if i not in tried]
        candidates = self.rng.sample(candidates, k=min(len(candidates), self.maxsteps))
        i = 0
        while i < len(candidates):
            cand = candidates[i]
            tried.add(cand)
            c_cluster, c_result, c_density = self._sample(cand)
            if c_density > local_density:
                medoid, result, local_density = cand, c_result, c_density
                candidates = [j for j in c_cluster.tolist() if j not in tried]
                candidates = self.rng.sample(candidates, k=min(len(candidates), self.maxsteps))
                i = 0
            else:
                i += 1
        return medoid, result

    # cluster.py:452-543
    def _threshold(self, result):
        if result["n_lt"] == 1:
            return "loner"
        dens = smooth(result["hist_ref"] if "hist_ref" in result else histogram_value(result["hist_fx"]))
        t = pick_threshold(dens, self.peak_valley_ratio)
        return "none" if t is None else t

    # cluster.py:545-604
    def _find_cluster(self):
        while True:
            seed = self._next_seed()
            medoid, result = self._wander(seed)
            thr = self._threshold(result)
            orig_medoid = int(self.indices[medoid])
            if thr == "loner":
                c = OracleCluster(orig_medoid, seed, np.array([orig_medoid]), self.peak_valley_ratio,
                                  None, None, self.successes, len(self.attempts))
                return c, np.array([medoid])
            if thr == "none":
                if self.peak_valley_ratio > 0.55:
                    points = select(self._live(), None, medoid, DEFAULT_RADIUS)
                    c = OracleCluster(orig_medoid, seed, self.indices[points], self.peak_valley_ratio,
                                      None, DEFAULT_RADIUS, self.successes, len(self.attempts))
                    return c, points
                self._update_successes(False)
                continue
            threshold, observed = thr
            points = select(self._live(), None, medoid, threshold)
            c = OracleCluster(orig_medoid, seed, self.indices[points], self.peak_valley_ratio, observed,
                              threshold, self.successes, len(self.attempts))
            if self.peak_valley_ratio < 0.55:
                self._update_successes(True)
            return c, points
