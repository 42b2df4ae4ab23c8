"""CPU test: the native cluster state machine's random.Random(seed).sample must be CPython's, draw for draw
(MT19937 init_by_array seeding, getrandbits, _randbelow_with_getrandbits, both branches of random.sample)."""
import random

import numpy as np
import pytest

from vamb_amd import _lib


@pytest.mark.parametrize("seed", [0, 1, 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 63 + 12345, 2 ** 64 - 1])
def test_sample_matches_cpython(seed):
    lib = _lib.load()
    calls = []
    # k <= 5: set size 21; k > 5: 21 + 4 ** ceil(log(3 k, 4)) -> n on both sides of every boundary
    for k in (0, 1, 2, 5, 6, 7, 21, 25):
        setsize = 21 + (4 ** int(np.ceil(np.log(k * 3) / np.log(4))) if k > 5 else 0)
        for n in (k, k + 1, 20, 21, 22, setsize - 1, setsize, setsize + 1, 300, 1000, 5000, 70000):
            if n >= k and n >= 1:
                calls.append((n, k))
    rng = random.Random(seed)
    want = []
    for n, k in calls:
        want.extend(rng.sample(range(n), k))
    ns = np.array([c[0] for c in calls], np.int64)
    ks = np.array([c[1] for c in calls], np.int64)
    out = np.empty(int(ks.sum()), np.int64)
    _lib.check(lib.vh_debug_pyrandom_sample(seed, len(calls), _lib.ptr(ns), _lib.ptr(ks), _lib.ptr(out)))
    assert out.tolist() == want


def test_sample_long_stream():
    """Many draws from one generator (several regenerations of the 624-word state)."""
    lib = _lib.load()
    rng = random.Random(7)
    calls = [(1 + (i * 37) % 900, min(25, 1 + (i * 37) % 900)) for i in range(400)]
    want = []
    for n, k in calls:
        want.extend(rng.sample(range(n), k))
    ns = np.array([c[0] for c in calls], np.int64)
    ks = np.array([c[1] for c in calls], np.int64)
    out = np.empty(int(ks.sum()), np.int64)
    _lib.check(lib.vh_debug_pyrandom_sample(7, len(calls), _lib.ptr(ns), _lib.ptr(ks), _lib.ptr(out)))
    assert out.tolist() == want


def test_native_find_threshold_matches_python():
    """The C++ find_threshold (float32 smoothing in the reference's summation order, double-precision peak / valley
    walk) against the Python implementation on randomly shaped histograms, for every peak_valley_ratio the success
    window can produce."""
    import ctypes

    from vamb_amd import cluster as vc

    lib = _lib.load()
    rng = np.random.RandomState(3)
    kinds = {0: 0, 1: 0, 2: 0}
    for trial in range(3000):
        shape = trial % 6
        x = np.arange(60)
        if shape == 0:
            h = rng.gamma(2.0, 50.0, 60)
        elif shape == 1:      # near peak, valley, far mass: the normal-cluster case
            c = rng.randint(2, 15); w = rng.uniform(1, 5)
            h = 4000 * np.exp(-0.5 * ((x - c) / w) ** 2) + rng.uniform(0, 30) * np.maximum(x - rng.randint(15, 40), 0) ** 1.5
        elif shape == 2:
            h = np.cumsum(rng.gamma(1.0, 20.0, 60))          # no near peak
        elif shape == 3:
            h = rng.gamma(0.3, 1000.0, 60) * (rng.random_sample(60) < 0.2)   # sparse spikes
        elif shape == 4:
            h = np.zeros(60); h[rng.randint(0, 60)] = rng.uniform(1, 1e6)
        else:
            c = rng.randint(0, 25)
            h = 1e5 * np.exp(-0.5 * ((x - c) / rng.uniform(0.5, 3)) ** 2) + rng.uniform(0, 2e4, 60)
        hist_fx = np.rint(h * 256.0).astype(np.int64)
        n_lt = int(rng.randint(1, 4))
        pvr = 0.1 * rng.randint(1, 8)
        st = vc.ScanStats(0, n_lt, n_lt, hist_fx)
        gen = vc.ClusterGenerator.__new__(vc.ClusterGenerator)
        gen.peak_valley_ratio = pvr
        want = vc.ClusterGenerator.find_threshold(gen, st)
        kind, thr, obs = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.vh_debug_find_threshold(_lib.ptr(hist_fx), n_lt, pvr, ctypes.byref(kind), ctypes.byref(thr),
                                               ctypes.byref(obs)))
        kinds[kind.value] += 1
        if isinstance(want, vc.Loner):
            assert kind.value == 0
        elif isinstance(want, vc.NoThreshold):
            assert kind.value == 1, (trial, thr.value)
        else:
            assert kind.value == 2 and thr.value == want[0] and obs.value == want[1], (trial, want, thr.value, obs.value)
    assert min(kinds.values()) > 50, kinds
