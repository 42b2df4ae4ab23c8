"""Diagnostic (GPU box): which scheduling variant of the bf16 step leaves different bits than the one-stream baseline?
Trains the model of tests/test_vae_gpu.py::test_step_scheduling_variants_are_bit_identical once per setting (several times for
the settings given more than once) and prints, per run, whether parameters / optimiser state / latents equal the baseline's.

    python tools/gpu/gpu_variant_bits.py [precision=bf16] [repeats=3]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ["VAMBHIP_PRECISION"] = prec
from vamb_amd import encode as ve, synth  # noqa: E402

KNOBS = ("VAMBHIP_FORK_EVENTS", "VAMBHIP_SINGLE_STREAM", "VAMBHIP_VAE_FORK_AT_LOSS", "VAMBHIP_VAE_FORK_PLAN",
         "VAMBHIP_VAE_PREFETCH_BATCH", "VAMBHIP_VAE_LOSS_DPP", "VAMBHIP_VAE_FUSED_SKINNY", "VAMBHIP_VAE_OPT_SPLIT",
         "VAMBHIP_VAE_FUSED_FINALIZE", "VAMBHIP_VAE_FORK_MODE")
n, S = 5000, 6
ab, tnf, lens, _ = synth.features(n, S, seed=11)


def run(setting):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(setting)
    dl = ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=512, destroy=True)
    vae = ve.VAE(S, seed=4)
    vae.trainmodel(dl, nepochs=4, batchsteps=[2])
    sd = {k: v.numpy().copy() for k, v in vae.state_dict().items()}
    return sd, vae.optimizer_state(), vae.encode(dl)


def same(a, b):
    return a[1] == b[1] and all(np.array_equal(a[0][k], b[0][k]) for k in a[0]) and np.array_equal(a[2], b[2])


for dpp in ("1",):
    base = run({"VAMBHIP_FORK_EVENTS": "1", "VAMBHIP_SINGLE_STREAM": "1", "VAMBHIP_VAE_LOSS_DPP": dpp})
    print(f"loss_dpp={dpp}: baseline (one stream) d = {base[1]['d']!r}", flush=True)
    for setting in ({"VAMBHIP_FORK_EVENTS": "1", "VAMBHIP_SINGLE_STREAM": "1"}, {}, {"VAMBHIP_VAE_PREFETCH_BATCH": "0"},
                    {"VAMBHIP_VAE_FORK_AT_LOSS": "0", "VAMBHIP_VAE_FORK_PLAN": "0", "VAMBHIP_VAE_PREFETCH_BATCH": "0"},
                    {"VAMBHIP_VAE_FORK_AT_LOSS": "0", "VAMBHIP_VAE_FORK_PLAN": "0"}, {"VAMBHIP_VAE_FORK_PLAN": "15"},
                    {"VAMBHIP_VAE_FORK_PLAN": "15", "VAMBHIP_VAE_PREFETCH_BATCH": "0"}, {"VAMBHIP_VAE_FUSED_SKINNY": "0"},
                    {"VAMBHIP_VAE_OPT_SPLIT": "1"}, {"VAMBHIP_VAE_OPT_SPLIT": "1", "VAMBHIP_VAE_PREFETCH_BATCH": "0"},
                    {"VAMBHIP_VAE_FUSED_FINALIZE": "1"}, {"VAMBHIP_VAE_FUSED_FINALIZE": "1", "VAMBHIP_VAE_PREFETCH_BATCH": "0"},
                    {"VAMBHIP_VAE_FUSED_SKINNY": "0", "VAMBHIP_VAE_FUSED_FINALIZE": "1", "VAMBHIP_SINGLE_STREAM": "1"},
                    {"VAMBHIP_VAE_FORK_MODE": "2"}, {"VAMBHIP_VAE_FORK_MODE": "2", "VAMBHIP_VAE_PREFETCH_BATCH": "0"}):
        res = []
        for _ in range(reps):
            r = run(dict(setting, VAMBHIP_VAE_LOSS_DPP=dpp))
            res.append("same" if same(base, r) else f"DIFF d={r[1]['d']!r}")
        print(f"  {setting}: {res}", flush=True)
