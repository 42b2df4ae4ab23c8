"""Diagnostic: per-tensor gradient / loss / latent errors of the bf16-operand path against the fp64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["VAMBHIP_PRECISION"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import fixture_defs as fd
import vae_oracle as vo
import test_vae_gpu as T

for name in ("vae_small_drop", "vae_default_arch"):
    c = fd.VAE_CASES[name]; g = fd.load(name); masks, eps = fd.vae_randomness(name); B = c["batch"]
    vae, st0 = T.make_vae(c, name); dl = T.loader_from(g, B); vae._ensure_dataset(dl)
    oracle = vo.OracleVAE(c["nsamples"], c["nhiddens"], c["nlatent"], c["alpha"], c["beta"], c["dropout"], state=st0)
    d, t, a, w = g["depths"], g["tnf"], g["total_abundance"], g["weights"]
    for step in range(c["steps"]):
        use_masks = masks[step] if c["dropout"] > 0 else None
        losses = vae.train_batch(np.arange(B), eps=eps[step], masks=use_masks)
        o = oracle.train_step(d[:B], t[:B], a[:B], w[:B], eps[step], masks[step])
        print(name, "step", step, "loss rel", T.rel(losses, o))
        if step == 0:
            for n in oracle.names:
                got = vae.parameters_gradient(n); ref = oracle.grads[n]
                print("   ", n, "max/max %.4f  fro %.4f" % (np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12),
                                                           np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)))
    lat = vae.encode(dl); ref = oracle.encode(d, t, a)
    print(name, "latent max err / max", np.abs(lat - ref).max() / np.abs(ref).max())
