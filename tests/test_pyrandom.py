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
