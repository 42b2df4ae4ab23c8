"""TEST INFRASTRUCTURE ONLY -- where does the gradient error of the bf16-storage training step come from?

An fp64 restatement of the DATAFLOW of vamb_amd's bf16 step (csrc/vae_step16.hpp: which tensors are stored as bf16, where the
BatchNorm statistics / bias gradients are summed) with a switch per rounding point, compared with the exact fp64 gradients of
``vae_oracle.OracleVAE`` on the same batch.  It answers, on the CPU, (a) how large the error is that "bf16 operands, fp32
accumulate" (BASELINE configs[2]) implies by itself and (b) which rounding points dominate, so that the GPU kernels spend
precision where it matters.  No HIP code is involved: this is the arithmetic model the GPU tolerances are derived from.

    python oracle/bf16_error_budget.py [batch] [nsamples] > profiles/r03_bf16_error_budget.txt
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import vae_oracle as vo  # noqa: E402
from vamb_amd import synth  # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


class Flow:
    """One training step in the GPU's dataflow.  ``r``: set of rounding points that are ON:
    x w h z dR dA dZ dMU  (tensor stored as bf16)   fstat bstat dbias  (sums taken from the ROUNDED tensor)."""

    def __init__(self, state, nsamples, nhiddens, nlatent, alpha, beta, dropout, r):
        self.st = {k: np.asarray(v, np.float64) for k, v in state.items()}
        self.S, self.hid, self.L = nsamples, list(nhiddens), nlatent
        self.alpha, self.beta, self.p = alpha, beta, dropout
        self.r = set(r)

    def q(self, name, x):
        return bf16(x) if name in self.r else x

    def step(self, depths, tnf, ab, w, eps, masks):
        st, q = self.st, self.q
        nl = len(self.hid)
        B = len(depths)
        scale = 1.0 / (1.0 - self.p) if self.p > 0 else 1.0
        x = np.concatenate([depths, tnf, ab], axis=1).astype(np.float64)
        layers = [(f"encoderlayers.{i}", f"encodernorms.{i}") for i in range(nl)] + \
                 [(f"decoderlayers.{i}", f"decodernorms.{i}") for i in range(nl)]
        tape = []
        a_in = q("x", x)
        prev = None   # (s, t) of the BatchNorm feeding the next Linear (folded into its weights)

        def folded(lin):
            W, b = st[lin + ".weight"], st[lin + ".bias"]
            if prev is None:
                return q("w", W), b
            s, t = prev
            return q("w", W * s[None, :]), b + W @ t

        for li, (lin, norm) in enumerate(layers):
            if li == nl:   # latent
                Wf, bf = folded("mu")
                mu = a_in @ Wf.T + bf
                tape_mu = dict(a_in=a_in, prev=prev)
                a_in = q("z", mu + eps)
                prev = None
            Wf, bf = folded(lin)
            z = a_in @ Wf.T + bf
            hfull = np.where(z > 0, z, 0.01 * z) * (masks[li] * scale if self.p > 0 else 1.0)
            h16 = q("h", hfull)
            hs = h16 if "fstat" in self.r else hfull
            mean = hs.mean(axis=0)
            var = np.maximum((hs * hs).mean(axis=0) - mean * mean, 0.0)
            istd = 1.0 / np.sqrt(var + vo.BN_EPS)
            s = istd * st[norm + ".weight"]
            t = st[norm + ".bias"] - mean * s
            tape.append(dict(lin=lin, norm=norm, a_in=a_in, prev=prev, z=z, h16=h16, mean=mean, istd=istd, s=s, t=t, mask=masks[li]))
            a_in, prev = h16, (s, t)
        Wf, bf = folded("outputlayer")
        recon = a_in @ Wf.T + bf
        # loss gradient (exact arithmetic; the kernel is fp32)
        ce_w, ab_w, sse_w, kld_w = vo.loss_weights(self.S, self.L, self.alpha, self.beta)
        S = self.S
        logits = recon[:, :S]
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        p = e / e.sum(axis=1, keepdims=True)
        g = w.reshape(-1).mean() / B
        dp = g * ce_w * (-depths / (p + 1e-9))
        dlogit = p * (dp - (p * dp).sum(axis=1, keepdims=True))
        drecon = np.concatenate([dlogit, g * sse_w * 2.0 * (recon[:, S:S + 103] - tnf), g * ab_w * 2.0 * (recon[:, S + 103:] - ab)], axis=1)
        dmu_kld = g * kld_w * mu
        grads = {}

        def complete(G, dbias, prev):   # dW of a layer whose input is normalised: G diag(s) + dbias t^T
            if prev is None:
                return G
            s, t = prev
            return G * s[None, :] + dbias[:, None] * t[None, :]

        dR = q("dR", drecon)
        last = tape[-1]
        dbo = dR.sum(axis=0)
        grads["outputlayer.weight"] = complete(dR.T @ last["h16"], dbo, (last["s"], last["t"]))
        grads["outputlayer.bias"] = dbo
        dA = dR @ q("w", st["outputlayer.weight"])

        def hidden_bwd(dA, t):
            dA16 = q("dA", dA)
            xhat = (t["h16"] - t["mean"]) * t["istd"]
            ds = dA16 if "bstat" in self.r else dA
            S1, S2 = ds.sum(axis=0), (ds * xhat).sum(axis=0)
            grads[t["norm"] + ".weight"] = S2
            grads[t["norm"] + ".bias"] = S1
            gam = st[t["norm"] + ".weight"]
            l = t["istd"] * gam * (dA16 - S1 / B - xhat * (S2 / B))
            keep = (t["mask"] * scale) if self.p > 0 else 1.0
            dz = l * keep * np.where(t["h16"] > 0, 1.0, 0.01)
            dZ = q("dZ", dz)
            db = (dZ if "dbias" in self.r else dz).sum(axis=0)
            grads[t["lin"] + ".bias"] = db
            grads[t["lin"] + ".weight"] = complete(dZ.T @ t["a_in"], db, t["prev"])
            return dZ @ q("w", st[t["lin"] + ".weight"])

        for li in range(2 * nl - 1, nl - 1, -1):
            dA = hidden_bwd(dA, tape[li])
        dMU = q("dMU", dA + dmu_kld)
        dbm = dMU.sum(axis=0)
        grads["mu.weight"] = complete(dMU.T @ tape_mu["a_in"], dbm, tape_mu["prev"])
        grads["mu.bias"] = dbm
        dA = dMU @ q("w", st["mu.weight"])
        for li in range(nl - 1, -1, -1):
            dA = hidden_bwd(dA, tape[li])
        return grads


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    hid, L, drop = [512, 512], 32, 0.2
    ab, tnf, lens, _ = synth.features(batch, S, seed=21)
    # the reference's normalisation (numpy statements of make_dataloader)
    a = ab * (1_000_000 / ab.sum(axis=0))
    tot = a.sum(axis=1)
    a = a / np.where(tot == 0, 1, tot)[:, None]
    la = np.log(np.clip(tot, 0.001, None)); la = ((la - la.mean()) / la.std())[:, None]
    tz = (tnf - tnf.mean(axis=0)) / np.where(tnf.std(axis=0) == 0, 1, tnf.std(axis=0))
    wts = np.maximum(np.log(lens.astype(np.float64)) - 5.0, 2.0); wts = (wts * len(wts) / wts.sum())[:, None]
    st0 = vo.init_state(S, hid, L, 5)
    rng = np.random.RandomState(1)
    eps = rng.standard_normal((batch, L))
    masks = [(rng.random_sample((batch, 512)) >= drop).astype(np.float64) for _ in range(4)]
    alpha = 0.15
    ora = vo.OracleVAE(S, hid, L, alpha, 200.0, drop, state=st0)
    ora.train_step(a, tz, la, wts, eps, masks)
    ref = ora.grads
    ALL = ["x", "w", "h", "z", "dR", "dA", "dZ", "dMU", "fstat", "bstat", "dbias"]
    configs = [("exact dataflow (no rounding)", []),
               ("round-2 GPU path: everything bf16, sums of rounded tensors", ALL),
               ("round-3: sums from fp32 (fstat, dbias), bstat still rounded", [x for x in ALL if x not in ("fstat", "dbias")]),
               ("all sums from fp32 (fstat, bstat, dbias)", [x for x in ALL if x not in ("fstat", "bstat", "dbias")]),
               ("... and dA kept fp32", [x for x in ALL if x not in ("fstat", "bstat", "dbias", "dA")]),
               ("only the GEMM operands of the FORWARD rounded (x w h z)", ["x", "w", "h", "z"]),
               ("only w", ["w"]), ("only h", ["h"]), ("only dR", ["dR"]), ("only dA (+ rounded sums)", ["dA", "bstat"]),
               ("only dA (sums fp32)", ["dA"]), ("only dZ (+ rounded dbias)", ["dZ", "dbias"]), ("only dZ (sums fp32)", ["dZ"])]
    names = vo.param_names(hid)
    print(f"batch {batch}, nsamples {S}, hidden {hid}, latent {L}, dropout {drop}: Frobenius error of every parameter gradient against fp64")
    print("%-64s %8s %8s %8s  worst tensor" % ("rounding points ON", "median", "mean", "max"))
    for label, r in configs:
        g = Flow(st0, S, hid, L, alpha, 200.0, drop, r).step(a, tz, la, wts, eps, masks)
        errs = {n: np.linalg.norm(g[n] - ref[n]) / max(np.linalg.norm(ref[n]), 1e-300) for n in names}
        v = np.array(list(errs.values()))
        worst = max(errs, key=errs.get)
        print("%-64s %8.2e %8.2e %8.2e  %s" % (label, np.median(v), v.mean(), v.max(), worst))
        if label.startswith(("round-2", "all sums", "... and")):
            for n in names:
                print("      %-26s %.2e" % (n, errs[n]))


if __name__ == "__main__":
    main()
