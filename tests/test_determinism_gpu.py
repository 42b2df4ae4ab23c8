"""Run-to-run determinism at size: the hand-scheduled kernels (counted ``s_waitcnt``, LDS-DMA rings, the
host-mapped result ring of the sweep, two-stream steps) are only correct if NOTHING depends on timing.  A race shows up as a pass
count, a cluster stream or a parameter bit that varies between identical runs -- round 5 found one that way by eye
(``profiles/r05t_sweep_prefill.txt``).  These tests make such a race fail the suite:

* the sweep of 200 000 x 32 sigma = 0.5 blobs (31 k+ clusters, every kind, speculation / lazy validation / compaction all active)
  three times: identical ``vh_gen_counters`` (passes, medoids) and identical stream hash;
* 500 consecutive training steps at the C2 shape (200 samples, batch 8192, default architecture, hash dropout, device-side
  shuffle and noise) twice from the same seed: bit-identical parameters, optimiser state and latents, fp32 and bf16;
* ``tests/diagnostics/gpu_determinism_step.py``'s one-step comparison (activations and every gradient) as a collected test.
"""
import hashlib

import numpy as np
import pytest

import fixture_defs as fd
import vae_oracle as vo
from vamb_amd import cluster as vc, encode as ve, synth

pytestmark = pytest.mark.gpu


def _stream_hash(clusters) -> str:
    h = hashlib.sha256()
    for c in clusters:
        h.update(np.int64(c.medoid).tobytes())
        h.update(np.asarray(c.members, dtype=np.int64).tobytes())
        h.update(np.float64(-1.0 if c.radius is None else c.radius).tobytes())
        h.update(c.kind_str.encode())
    return h.hexdigest()


def test_sweep_counters_and_stream_are_identical_run_to_run():
    lat, _ = synth.blob_latent(200_000, 32, sigma=0.5, seed=17)
    lens = synth.lengths(200_000, 17)
    runs = []
    for _ in range(3):
        gen = vc.ClusterGenerator(lat.copy(), lens, destroy=True, rng_seed=0)
        clusters = list(gen)
        gen._sync_native_counters()
        b = gen._backend
        runs.append((len(clusters), int(b.scan_passes), int(b.scan_medoids), _stream_hash(clusters)))
        b.close()
    assert runs[0][0] > 10_000 and runs[0][1] > 10_000 and runs[0][2] > runs[0][1]   # a real sweep: tens of thousands of passes
    assert runs[1] == runs[0] and runs[2] == runs[0], runs


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_500_training_steps_at_the_c2_shape_are_bit_identical(dtype, monkeypatch):
    monkeypatch.setenv("VAMBHIP_PRECISION", dtype)
    S, B, steps_per_epoch, epochs = 200, 8192, 25, 20            # 500 steps of the benchmarked step
    n = B * steps_per_epoch
    ab, tnf, lens, _ = synth.features(n, S, seed=31)
    out = []
    for _ in range(2):
        dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B, destroy=True)
        vae = ve.VAE(S, seed=9)
        vae.trainmodel(dl, nepochs=epochs, batchsteps=None)
        out.append(({k: v.numpy().copy() for k, v in vae.state_dict().items()}, vae.optimizer_state(),
                    vae.encode(dl)[:4096].copy(), dict(vae.last_epoch_losses)))
        del vae, dl
    (a, oa, la, qa), (b, ob, lb, qb) = out
    assert oa == ob and qa == qb
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(la, lb) and np.isfinite(la).all()


@pytest.mark.parametrize("dtype,batch,S", [("fp32", 512, 6), ("bf16", 512, 6), ("bf16", 8192, 200)])
def test_one_free_running_step_is_bit_identical(dtype, batch, S, monkeypatch):
    """(tests/diagnostics/gpu_determinism_step.py, collected) the losses, the activations of every hidden layer and every
    gradient tensor of ONE free-running step on identical fresh models, three times."""
    monkeypatch.setenv("VAMBHIP_PRECISION", dtype)
    ab, tnf, lens, _ = synth.features(max(batch, 2048), S, seed=11)
    names = vo.param_names([512, 512])
    runs = []
    for _ in range(3):
        dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=batch, destroy=True)
        vae = ve.VAE(S, seed=4)
        vae._ensure_dataset(dl)
        rec = {"losses": np.array(vae.train_batch(np.arange(batch)))}
        for li in range(4):
            rec[f"hidden{li}"] = vae.hidden_activations(li, batch)
        for nm in names:
            rec["grad:" + nm] = vae.parameters_gradient(nm)
        runs.append(rec)
    for k in runs[0]:
        for r in runs[1:]:
            assert np.array_equal(r[k], runs[0][k]), k
