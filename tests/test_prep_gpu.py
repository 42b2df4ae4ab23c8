"""make_dataloader on the device (csrc/prep.hip) against the host numpy path, which is the
reference's own statements (vamb/encode.py:98-126) and is itself pinned to tensors produced by the real reference
(tests/test_prep_host.py, tests/golden/prep_*.npz).  Bar: bit-exact float32 tensors."""
import numpy as np
import pytest
import torch

import fixture_defs as fd
from vamb_amd import encode as ve, synth

pytestmark = pytest.mark.gpu


def _both(ab, tnf, lens, batchsize=64):
    ve.set_prep_mode("host")
    try:
        host = [t.numpy() for t in ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=batchsize).dataset.tensors]
        ve.set_prep_mode("device")
        dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=batchsize)
        assert getattr(dl.dataset, "_vambhip_prepared", None) is not None     # the device path really ran
        dev = [t.numpy() for t in dl.dataset.tensors]
    finally:
        ve.set_prep_mode("auto")
    return host, dev, dl


@pytest.mark.parametrize("name", list(fd.PREP_CASES))
def test_device_prep_matches_reference_tensors(name):
    ab, tnf, lens = fd.prep_inputs(name)
    g = fd.load(name)
    host, dev, _ = _both(ab, tnf, lens, 16)
    for h, d, key in zip(host, dev, ("depths", "tnf", "total_abundance", "weights")):
        assert d.dtype == np.float32 and d.shape == g[key].shape
        assert np.array_equal(h, d), key          # device == host numpy path on this machine, bit for bit
        if key in ("depths", "tnf"):
            assert np.array_equal(d, g[key]), key
        else:
            # total_abundance and weights go through numpy's float32 log, whose SIMD kernel is chosen per host CPU
            # (AVX512F / AVX2 dispatch): robust against a last-ulp difference between the golden's CPU and this host
            assert np.allclose(d, g[key], rtol=2e-6, atol=2e-6), key


@pytest.mark.parametrize("n,S", [(1, 1), (100, 1), (5000, 1), (2, 3), (600, 2), (33, 7), (100, 8), (513, 9), (1000, 50), (4097, 129), (3000, 200),
                                 (700, 1000), (257, 2500), (20011, 137)])
def test_device_prep_bit_exact(n, S):
    ab, tnf, lens, _ = synth.features(n, S, seed=n + S)
    rng = np.random.RandomState(n)
    if n > 10:      # contigs with zero depth everywhere take the 1 / n_samples branch (encode.py:109-113)
        ab[rng.choice(n, size=max(1, n // 50), replace=False)] = 0.0
    if n > 4:
        tnf[:, 5] = tnf[0, 5]        # a constant TNF column: std == 0 -> 1 (vambtools.py:276)
    host, dev, dl = _both(ab, tnf, lens)
    for h, d, key in zip(host, dev, ("depths", "tnf", "total_abundance", "weights")):
        assert h.shape == d.shape and np.array_equal(h, d, equal_nan=True), key
    assert len(dl.dataset) == n and len(dl.dataset[0]) == 4


def test_device_prep_full_size_properties():
    """C2-sized input (2 M x 200): too large for an element-wise host comparison inside the test budget, so the
    size-independent properties: rows of depths sum to 1, TNF columns have zero mean / unit variance, and a 64 k-row
    prefix block equals the host path when the column statistics of the full matrix are reused (checked through the
    first rows of the tensors only where they do not depend on the other rows: depths / total given the same scale)."""
    n, S = 2_000_000, 200
    ab, tnf, lens, _ = synth.features(n, S, seed=5)
    ve.set_prep_mode("device")
    try:
        dl = ve.make_dataloader(ab, tnf, lens, batchsize=8192)
    finally:
        ve.set_prep_mode("auto")
    d, t, a, w = (x.numpy() for x in dl.dataset.tensors)
    assert np.abs(d.sum(axis=1, dtype=np.float64) - 1).max() < 1e-5
    # (float32 row-order sums over 2 M rows, as numpy computes them: percent-level agreement with a float64 recomputation)
    assert np.abs(t.mean(axis=0, dtype=np.float64)).max() < 5e-2 and np.abs(t.std(axis=0, dtype=np.float64) - 1).max() < 5e-2
    # the depth block depends on the other rows only through the column sums: recompute it on the host for a prefix
    sums = ab.sum(axis=0)
    m = 65536
    x = ab[:m] * (1_000_000 / sums)
    tot = x.sum(axis=1)
    assert np.array_equal(d[:m], x / tot.reshape(-1, 1))
    assert abs(float(w.sum()) - n) < 1.0


def test_zero_depth_sample_raises():
    ab, tnf, lens, _ = synth.features(100, 6, seed=1)
    ab[:, 2] = 0
    ve.set_prep_mode("device")
    try:
        with pytest.raises(ValueError):
            ve.make_dataloader(ab, tnf, lens, batchsize=32)
    finally:
        ve.set_prep_mode("auto")


def test_training_from_a_prepared_loader_is_identical():
    """The VAE consumes the device-prepared matrix directly (no host round trip): same seed -> same losses and the same
    latents as training on the host-prepared loader."""
    ab, tnf, lens, _ = synth.features(3000, 20, seed=2)
    out = []
    for mode in ("host", "device"):
        ve.set_prep_mode(mode)
        try:
            dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=256)
        finally:
            ve.set_prep_mode("auto")
        vae = ve.VAE(20, seed=3)
        vae.trainmodel(dl, nepochs=3, batchsteps=None)
        out.append((vae.last_epoch_losses["loss"], vae.encode(dl)))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])


# ---- the reference's own make_dataloader tests (test/test_encode.py:47-95) with a GPU visible and prep mode "auto" ---------
def _ref_fixture():
    rng = np.random.RandomState(0)
    tnfs = rng.random_sample((111, 103)).astype(np.float32)
    rpkm = rng.random_sample((111, 14)).astype(np.float32)
    lens = rng.randint(2000, 5000, size=111)
    return tnfs, rpkm, lens


def _nearly_same(a, b):
    return bool(np.all(np.abs(a - b) < 1e-5))


def test_reference_test_destroy_on_the_device_path():
    """test_encode.py:47-58 (test_destroy): destroy=False leaves the caller's arrays alone, destroy=True normalises them in
    place -- on the DEVICE path (prep mode auto, GPU visible), where the normalised blocks are written back."""
    assert ve.get_prep_mode() == "auto"
    tnfs, rpkm, lens = _ref_fixture()
    copy_rpkm, copy_tnfs = rpkm.copy(), tnfs.copy()
    dl = ve.make_dataloader(rpkm, tnfs, lens, batchsize=32)
    assert getattr(dl.dataset, "_vambhip_prepared", None) is not None
    assert _nearly_same(rpkm, copy_rpkm) and _nearly_same(tnfs, copy_tnfs)
    dl = ve.make_dataloader(copy_rpkm, copy_tnfs, lens, batchsize=32, destroy=True)
    assert getattr(dl.dataset, "_vambhip_prepared", None) is not None
    assert not _nearly_same(rpkm, copy_rpkm) and not _nearly_same(tnfs, copy_tnfs)
    # the dataset's host tensors ARE the caller's arrays (torch.from_numpy in the reference, encode.py:128-133)
    assert dl.dataset.tensors[0].numpy().ctypes.data == copy_rpkm.ctypes.data
    assert dl.dataset.tensors[1].numpy().ctypes.data == copy_tnfs.ctypes.data
    # and they hold exactly what the host path writes into them
    ve.set_prep_mode("host")
    try:
        h_rpkm, h_tnfs = rpkm.copy(), tnfs.copy()
        ve.make_dataloader(h_rpkm, h_tnfs, lens, batchsize=32, destroy=True)
    finally:
        ve.set_prep_mode("auto")
    assert np.array_equal(h_rpkm, copy_rpkm) and np.array_equal(h_tnfs, copy_tnfs)


def test_reference_test_normalized_on_the_device_path():
    """test_encode.py:60-75 (test_normalized)."""
    tnfs, rpkm, lens = _ref_fixture()
    copy_rpkm, copy_tnfs = rpkm.copy(), tnfs.copy()
    ve.make_dataloader(copy_rpkm, copy_tnfs, lens, batchsize=32, destroy=True)
    assert _nearly_same(np.mean(copy_tnfs, axis=0), np.zeros(copy_tnfs.shape[1]))
    assert _nearly_same(np.std(copy_tnfs, axis=0), np.ones(copy_tnfs.shape[1]))
    assert _nearly_same(np.sum(copy_rpkm, axis=1), np.ones(copy_rpkm.shape[0]))
    assert np.all(copy_rpkm >= 0.0)


def test_reference_test_single_sample_on_the_device_path():
    """test_encode.py:77-95 (test_single_sample)."""
    tnfs, rpkm, lens = _ref_fixture()
    single_rpkm = rpkm[:, [0]]
    copy_single = single_rpkm.copy()
    dl = ve.make_dataloader(single_rpkm, tnfs.copy(), lens, batchsize=32, destroy=True)
    assert getattr(dl.dataset, "_vambhip_prepared", None) is not None
    assert abs(abs(np.mean(single_rpkm)) - 1.0) < 1e-6
    assert abs(np.std(single_rpkm)) < 1e-6
    assert (torch.argsort(dl.dataset.tensors[2], dim=0, stable=True)
            == torch.argsort(torch.from_numpy(copy_single), dim=0, stable=True)).all().item()
