"""Deterministic synthetic inputs for the hot path.

Every generator is a pure function of ``(shape, seed)`` through ``numpy.random.RandomState`` so that
tests, golden fixtures (``tests/golden/make_golden.py``) and ``bench.py`` see identical bytes on any
machine.  Shapes follow the reference's inputs: abundance ``[N, S]`` float32, TNF ``[N, 103]`` float32
(low 12 mantissa bits cleared, as ``vamb/parsecontigs.py:211`` does), lengths ``[N]`` int64 >= 2000
(``vamb/__main__.py:2217``), latent ``[N, L]`` float32 (low 12 bits cleared, ``vamb/encode.py:483``).
"""
from __future__ import annotations

import numpy as np

NTNF = 103


def mask_lower_bits(floats: np.ndarray, bits: int = 12) -> None:
    """In-place clear of the low mantissa bits (same contract as vamb/vambtools.py:324-330)."""
    if bits < 0 or bits > 23:
        raise ValueError("Must mask between 0 and 23 bits")
    u = floats.view(np.uint32)
    u &= ~np.uint32(2 ** bits - 1)


def n_genomes(n: int) -> int:
    return max(10, n // 200)


def lengths(n: int, seed: int) -> np.ndarray:
    rng = np.random.RandomState(seed + 7919)
    v = 2000.0 + rng.lognormal(8.0, 1.0, size=n)
    return np.clip(v, 2000, 1_000_000).astype(np.int64)


def blob_latent(n: int, nlatent: int = 32, sigma: float = 0.08, seed: int = 0, k: int | None = None):
    """Gaussian-blob latents: genome centres N(0, I), members centre + sigma * eps.

    Returns (latent float32 [n, nlatent], labels int64 [n]).  sigma=0.08 gives ~100 % "normal" clusters,
    sigma=0.5 exercises loner / NoThreshold / fallback / PVR relaxation (SURVEY.md 8d probe).
    """
    rng = np.random.RandomState(seed)
    k = n_genomes(n) if k is None else k
    centres = rng.standard_normal((k, nlatent))
    labels = rng.randint(0, k, size=n)
    lat = centres[labels] + sigma * rng.standard_normal((n, nlatent))
    lat = np.ascontiguousarray(lat.astype(np.float32))
    mask_lower_bits(lat, 12)
    return lat, labels.astype(np.int64)


def features(n: int, nsamples: int, seed: int = 0, k: int | None = None, chunk: int = 262144):
    """Raw (un-normalised) abundance [n, S], tnf [n, 103], lengths [n] and genome labels.

    Generated in row chunks so the 2M x 1000 configuration never holds a float64 copy of the matrix.
    """
    rng = np.random.RandomState(seed)
    k = n_genomes(n) if k is None else k
    a_g = rng.standard_normal((k, nsamples)).astype(np.float32)
    t_g = (0.1 * rng.standard_normal((k, NTNF))).astype(np.float32)
    labels = rng.randint(0, k, size=n)
    abundance = np.empty((n, nsamples), np.float32)
    tnf = np.empty((n, NTNF), np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        lab = labels[lo:hi]
        depth = rng.lognormal(0.0, 1.0, size=(hi - lo, 1)).astype(np.float32)
        a = np.exp(a_g[lab] + np.float32(0.3) * rng.standard_normal((hi - lo, nsamples)).astype(np.float32))
        a *= depth
        a[rng.random_sample((hi - lo, nsamples)) < 0.01] = 0.0
        abundance[lo:hi] = a
        tnf[lo:hi] = t_g[lab] + np.float32(0.05) * rng.standard_normal((hi - lo, NTNF)).astype(np.float32)
    # a sample with zero depth everywhere is rejected by make_dataloader (encode.py:99-103)
    dead = np.flatnonzero(abundance.sum(axis=0) == 0)
    abundance[0, dead] = 1.0
    mask_lower_bits(tnf, 12)
    return abundance, tnf, lengths(n, seed), labels.astype(np.int64)
