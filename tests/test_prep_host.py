"""vamb_amd.encode.make_dataloader / set_batchsize (host numpy, reference encode.py:33-146) against
tensors produced by the REAL reference (tests/golden/prep_*.npz) and the reference's own unit-test
properties (test/test_encode.py:8-119)."""
import numpy as np
import pytest
import torch

import fixture_defs as fd
from vamb_amd import encode as ve


@pytest.mark.parametrize("name", list(fd.PREP_CASES))
def test_matches_reference_tensors(name):
    ab, tnf, lens = fd.prep_inputs(name)
    dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=16)
    g = fd.load(name)
    got = [t.numpy() for t in dl.dataset.tensors]
    for arr, key in zip(got, ("depths", "tnf", "total_abundance", "weights")):
        assert arr.dtype == np.float32 and arr.shape == g[key].shape
        assert np.array_equal(arr, g[key]), key   # same numpy calls in the same order: bit-exact


def test_bad_args():
    tnfs = np.random.random((111, 103)).astype(np.float32)
    rpkm = np.random.random((111, 14)).astype(np.float32)
    lens = np.random.randint(2000, 5000, size=111)
    with pytest.raises(ValueError):
        ve.make_dataloader([[1, 2, 3]], tnfs, lens, batchsize=32)
    with pytest.raises(ValueError):
        ve.make_dataloader(rpkm, [[1, 2, 3]], lens, batchsize=32)
    with pytest.raises(ValueError):
        ve.make_dataloader(rpkm, tnfs, lens, batchsize=0)
    with pytest.raises(ValueError):
        ve.make_dataloader(np.random.random((110,)).astype(np.float32), tnfs, lens, batchsize=32)
    with pytest.raises(ValueError):
        ve.make_dataloader(rpkm.astype(np.float64), tnfs, lens, batchsize=32)
    z = rpkm.copy()
    z[:, 3] = 0
    with pytest.raises(ValueError):
        ve.make_dataloader(z, tnfs, lens, batchsize=32)


def test_destroy_normalisation_and_iteration():
    rng = np.random.RandomState(0)
    tnfs = rng.random_sample((111, 103)).astype(np.float32)
    rpkm = rng.random_sample((111, 14)).astype(np.float32)
    lens = rng.randint(2000, 5000, size=111)
    c_r, c_t = rpkm.copy(), tnfs.copy()
    ve.make_dataloader(rpkm, tnfs, lens, batchsize=32)
    assert np.array_equal(rpkm, c_r) and np.array_equal(tnfs, c_t)
    dl = ve.make_dataloader(c_r, c_t, lens, batchsize=32, destroy=True)
    assert np.any(np.abs(rpkm - c_r) > 1e-4) and np.any(np.abs(tnfs - c_t) > 1e-4)
    assert np.all(np.abs(c_t.mean(axis=0)) < 1e-5) and np.all(np.abs(c_t.std(axis=0) - 1) < 1e-5)
    assert np.all(np.abs(c_r.sum(axis=1) - 1) < 1e-5) and np.all(c_r >= 0)
    batch = next(iter(dl))
    assert len(batch) == 4
    for m in batch:
        assert m.dtype == torch.float32 and m.shape[0] == 32
    assert len(dl) == 111 // 32 and dl.batch_size == 32
    enc = ve.set_batchsize(dl, 64, 111, encode=True)
    assert len(enc) == 2 and enc.batch_size == 64
    first = next(iter(enc))[1].numpy()
    assert np.array_equal(first, c_t[:64])          # ordered, not shuffled
    dbl = ve.set_batchsize(dl, 64, 111)
    assert len(dbl) == 1                             # drop_last


def test_single_sample():
    rng = np.random.RandomState(1)
    tnfs = rng.random_sample((111, 103)).astype(np.float32)
    single = rng.random_sample((111, 1)).astype(np.float32)
    cp = single.copy()
    dl = ve.make_dataloader(single, tnfs, rng.randint(2000, 5000, size=111), batchsize=32, destroy=True)
    assert abs(abs(single.mean()) - 1.0) < 1e-6 and abs(single.std()) < 1e-6
    assert (torch.argsort(dl.dataset.tensors[2], dim=0, stable=True)
            == torch.argsort(torch.from_numpy(cp), dim=0, stable=True)).all().item()


def test_device_shuffle_formula_is_a_permutation():
    """Python mirror of vae_kernels.hpp::shuffle_index (the device-side epoch shuffle): for every n and
    key the map restricted to [0, n) by cycle walking must be a bijection."""
    M = (1 << 64) - 1

    def rnd(x, key, mask, bits):
        s1, s2 = (bits + 1) // 2, max((bits + 2) // 3, 1)
        x = (x * 0x9E3779B97F4A7C15 + key) & M & mask
        x ^= x >> s1
        x = (x * 0xBF58476D1CE4E5B9 + (key >> 17)) & M & mask
        x ^= x >> s2
        x = (x * 0x94D049BB133111EB + (key >> 31)) & M & mask
        x ^= x >> s1
        return x

    for n in (1, 2, 3, 7, 111, 1000, 4097):
        bits = 1
        while (1 << bits) < n:
            bits += 1
        mask = (1 << bits) - 1
        for key in (1, 0x1234567890ABCDEF, 2 ** 63 + 12345):
            out = []
            for i in range(n):
                x = i
                while True:
                    x = rnd(x, key, mask, bits)
                    if x < n:
                        break
                out.append(x)
            assert sorted(out) == list(range(n)), (n, key)
            if n >= 1000:   # not the identity, and different keys give different orders
                assert sum(1 for i, v in enumerate(out) if i == v) < n // 50


# ---- the summation orders csrc/prep.hip reproduces (device side of make_dataloader) ----------
def _run_pairwise_program(program, row):
    """The postfix program of ve._pairwise_program executed in explicit float32 steps (what prep_rows_kernel does)."""
    f = np.float32
    stack = []
    for kind, start, length in program:
        if kind == 1:
            right, left = stack.pop(), stack.pop()
            stack.append(f(left + right))
            continue
        a = row[start:start + length]
        if length < 8:
            res = f(0.0)
            for v in a:
                res = f(res + v)
        else:
            body = length - length % 8
            r = [f(a[j]) for j in range(8)]
            for i in range(8, body, 8):
                for j in range(8):
                    r[j] = f(r[j] + a[i + j])
            res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
            for i in range(body, length):
                res = f(res + a[i])
        stack.append(res)
    assert len(stack) == 1
    return f(f(0.0) + stack[0])


@pytest.mark.parametrize("S", list(range(1, 20)) + [50, 127, 128, 129, 136, 137, 200, 255, 256, 257, 300, 1000, 1001, 2500])
def test_pairwise_program_is_numpys_row_sum(S):
    rng = np.random.RandomState(S)
    a = np.exp(3 * rng.standard_normal((5, S))).astype(np.float32)
    program = ve._pairwise_program(S)
    got = np.array([_run_pairwise_program(program, r) for r in a], dtype=np.float32)
    assert np.array_equal(got, a.sum(axis=1))


def test_column_sums_are_sequential_in_row_order():
    """a.sum(axis=0) / a.mean(axis=0) / a.std(axis=0) of a C-contiguous float32 matrix: one chain per column, rows
    ascending; the mean and variance divide in double and round once (numpy _mean / _var with an intp count)."""
    rng = np.random.RandomState(3)
    # (a single column is a contiguous 1-d reduction: pairwise, not row order -- the device path leaves that case to numpy)
    one = np.exp(rng.standard_normal((70001, 1))).astype(np.float32)
    assert np.array_equal(one.sum(axis=0), np.array([one[:, 0].sum()], np.float32))
    for n, c in [(1000, 2), (1000, 7), (4097, 103), (70001, 16)]:
        a = np.exp(rng.standard_normal((n, c))).astype(np.float32)
        acc = a[0].copy()
        for i in range(1, n):
            acc += a[i]
        assert np.array_equal(acc, a.sum(axis=0))
        mean = np.true_divide(acc.copy(), np.intp(n), casting="unsafe", out=acc.copy())
        assert np.array_equal(mean, a.mean(axis=0))
        x = a - mean
        x = x * x
        sq = x[0].copy()
        for i in range(1, n):
            sq += x[i]
        var = np.true_divide(sq, np.intp(n), out=sq, casting="unsafe")
        assert np.array_equal(np.sqrt(var), a.std(axis=0))
