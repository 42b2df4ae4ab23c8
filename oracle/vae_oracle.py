"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``vamb.encode.VAE`` numerics (explicit backward).

Part of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this.  The product (``vamb_amd/``) never does.

What is restated (citations into ``/root/reference/vamb/encode.py``):
  * ``VAE._encode`` / ``reparameterize`` / ``_decode`` / ``forward``      259-314
  * ``VAE.calc_loss``                                                    316-357
  * autograd of the above (hand-derived; checked against torch autograd run on the reference
    module itself -- golden vectors in ``tests/golden/vae_*.npz``)
  * ``torch.nn.BatchNorm1d`` train/eval semantics (eps 1e-5, momentum 0.1, biased variance for the
    normalisation, unbiased for the running estimate)                    238,246
  * ``dadaptation.DAdaptAdam.step`` (dadaptation==3.2; PARITY UNPINNED, see dadapt_restated.py)  578
  * ``VAE.encode`` + ``vambtools.mask_lower_bits(latent, 12)``            442-484

Randomness (dropout masks, reparameterisation noise, shuffle order) is always *injected* by the
caller: the reference draws it from torch's global generators (encode.py:210,277), which no other
implementation can reproduce, and says so itself (doc/how_to_run.md:108, test_results.py:11-15).

All arithmetic runs in ``dtype`` (float64 by default: an independent, more accurate truth for both
torch-fp32 and the HIP fp32 kernels; pass float32 to mimic fp32 rounding).
"""
from __future__ import annotations

from math import log

import numpy as np

NTNF = 103
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.01


def layer_dims(nsamples, nhiddens, nlatent, width=None):
    d = nsamples + NTNF + 1 if width is None else width   # width: the subclasses' input columns (semisup_oracle.py)
    enc = list(zip([d] + list(nhiddens), nhiddens))
    dec = list(zip([nlatent] + list(nhiddens[::-1]), nhiddens[::-1]))
    return d, enc, dec


def param_names(nhiddens):
    """state_dict order of the reference module (encode.py:226-249)."""
    names = []
    nl = len(nhiddens)
    for i in range(nl):
        names += [f"encoderlayers.{i}.weight", f"encoderlayers.{i}.bias"]
    for i in range(nl):
        names += [f"encodernorms.{i}.weight", f"encodernorms.{i}.bias"]
    for i in range(nl):
        names += [f"decoderlayers.{i}.weight", f"decoderlayers.{i}.bias"]
    for i in range(nl):
        names += [f"decodernorms.{i}.weight", f"decodernorms.{i}.bias"]
    names += ["mu.weight", "mu.bias", "outputlayer.weight", "outputlayer.bias"]
    return names


def init_state(nsamples, nhiddens, nlatent, seed, width=None):
    """Deterministic numpy initialisation with torch's default Linear scheme
    (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias; BN weight 1, bias 0, rm 0, rv 1).
    Used by fixtures so that weights never have to be stored."""
    rng = np.random.RandomState(seed)
    d, enc, dec = layer_dims(nsamples, nhiddens, nlatent, width)
    st = {}

    def lin(prefix, nin, nout):
        bound = 1.0 / np.sqrt(nin)
        st[prefix + ".weight"] = rng.uniform(-bound, bound, size=(nout, nin)).astype(np.float32)
        st[prefix + ".bias"] = rng.uniform(-bound, bound, size=(nout,)).astype(np.float32)

    def bn(prefix, n):
        # non-trivial affine so the gamma/beta gradients are exercised
        st[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        st[prefix + ".bias"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        st[prefix + ".running_mean"] = np.zeros(n, np.float32)
        st[prefix + ".running_var"] = np.ones(n, np.float32)
        st[prefix + ".num_batches_tracked"] = np.zeros((), np.int64)

    for i, (nin, nout) in enumerate(enc):
        lin(f"encoderlayers.{i}", nin, nout)
        bn(f"encodernorms.{i}", nout)
    lin("mu", nhiddens[-1], nlatent)
    for i, (nin, nout) in enumerate(dec):
        lin(f"decoderlayers.{i}", nin, nout)
        bn(f"decodernorms.{i}", nout)
    lin("outputlayer", nhiddens[0], d)
    return st


def loss_weights(nsamples, nlatent, alpha, beta):
    """encode.py:333-343"""
    ce_w = 0.0 if nsamples == 1 else ((1 - alpha) * (nsamples - 1)) / (nsamples * log(nsamples))
    ab_w = (1 - alpha) * (1 / nsamples)
    sse_w = alpha / NTNF
    kld_w = 1 / (nlatent * beta)
    return ce_w, ab_w, sse_w, kld_w


class OracleVAE:
    def __init__(self, nsamples, nhiddens=None, nlatent=32, alpha=None, beta=200.0, dropout=0.2,
                 state=None, dtype=np.float64, bn_sync=None):
        if alpha is None:
            alpha = 0.15 if nsamples > 1 else 0.50
        if nhiddens is None:
            nhiddens = [512, 512] if nsamples > 1 else [256, 256]
        if dropout is None:
            dropout = 0.2 if nsamples > 1 else 0.0
        self.nsamples, self.nhiddens, self.nlatent = nsamples, list(nhiddens), nlatent
        self.alpha, self.beta, self.dropout = alpha, beta, dropout
        self.dtype = dtype
        # data-parallel shard with synchronised BatchNorm: bn_sync(array) -> element-wise SUM of the array over all
        # ranks.  The batch statistics (and the two sums of the BatchNorm backward) then span the all-rank batch, which
        # is what the single-process reference computes on the whole batch (encode.py:238,246,264).
        self.bn_sync = bn_sync
        self.state = {k: (np.array(v, dtype=dtype) if v.dtype.kind == "f" else np.array(v))
                      for k, v in state.items()}
        self.names = param_names(self.nhiddens)
        # optimizer state (DAdaptAdam defaults used by Vamb)
        self.d = 1e-6
        self.k = 0
        self.numerator_weighted = 0.0
        self.opt = {n: dict(s=np.zeros_like(self.state[n]), m=np.zeros_like(self.state[n]),
                            v=np.zeros_like(self.state[n])) for n in self.names}

    # ---- forward pieces -------------------------------------------------------------------------
    def _hidden_fwd(self, a_prev, lin, norm, mask, train):
        st = self.state
        z = a_prev @ st[lin + ".weight"].T + st[lin + ".bias"]
        r = np.where(z > 0, z, LRELU_SLOPE * z)
        if train and self.dropout > 0:
            scale = self.dtype(1.0) / (self.dtype(1.0) - self.dtype(self.dropout))
            h = r * (mask.astype(self.dtype) * scale)
        else:
            h = r
        n_glob = h.shape[0]
        if train and self.bn_sync is not None:
            tot = self.bn_sync(np.concatenate([h.sum(axis=0), (h * h).sum(axis=0), [float(h.shape[0])]]))
            n_glob = int(round(tot[-1]))
            nc = h.shape[1]
            mean = tot[:nc] / n_glob
            var = np.maximum(tot[nc:2 * nc] / n_glob - mean * mean, 0.0)
            n = n_glob
        elif train:
            mean = h.mean(axis=0)
            var = ((h - mean) ** 2).mean(axis=0)
            n = h.shape[0]
        if train:
            st[norm + ".running_mean"] = (1 - BN_MOMENTUM) * st[norm + ".running_mean"] + BN_MOMENTUM * mean
            st[norm + ".running_var"] = (1 - BN_MOMENTUM) * st[norm + ".running_var"] + BN_MOMENTUM * var * (n / (n - 1))
            st[norm + ".num_batches_tracked"] = st[norm + ".num_batches_tracked"] + 1
        else:
            mean, var = st[norm + ".running_mean"], st[norm + ".running_var"]
        invstd = 1.0 / np.sqrt(var + BN_EPS)
        xhat = (h - mean) * invstd
        a = xhat * st[norm + ".weight"] + st[norm + ".bias"]
        return a, dict(a_prev=a_prev, z=z, mask=mask, xhat=xhat, invstd=invstd, lin=lin, norm=norm, n_glob=n_glob)

    def forward(self, depths, tnf, abundance, eps=None, masks=None, train=True):
        """encode.py:306-314.  ``masks``: list of 2*len(nhiddens) boolean [B, n] keep-masks in the
        order the reference applies dropout (encoder layers, then decoder layers).  ``eps``: [B, L]."""
        dt = self.dtype
        x = np.concatenate([depths, tnf, abundance], axis=1).astype(dt)
        nl = len(self.nhiddens)
        tape = []
        a = x
        for i in range(nl):
            a, t = self._hidden_fwd(a, f"encoderlayers.{i}", f"encodernorms.{i}",
                                    None if masks is None else masks[i], train)
            tape.append(t)
        mu = a @ self.state["mu.weight"].T + self.state["mu.bias"]
        lat = mu + (0 if eps is None else eps.astype(dt))
        tape_mu = dict(a_prev=a)
        a = lat
        for i in range(nl):
            a, t = self._hidden_fwd(a, f"decoderlayers.{i}", f"decodernorms.{i}",
                                    None if masks is None else masks[nl + i], train)
            tape.append(t)
        recon = a @ self.state["outputlayer.weight"].T + self.state["outputlayer.bias"]
        S = self.nsamples
        logits = recon[:, :S]
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        depths_out = e / e.sum(axis=1, keepdims=True)
        tnf_out = recon[:, S:S + NTNF]
        ab_out = recon[:, S + NTNF:]
        self._tape = dict(hidden=tape, mu=tape_mu, a_last=a, x=x)
        return depths_out, tnf_out, ab_out, mu

    def calc_loss(self, depths_in, depths_out, tnf_in, tnf_out, ab_in, ab_out, mu, weights,
                  global_wsum=None, global_batch=None):
        """encode.py:316-357.  Returns the 5 scalars and stashes what backward needs."""
        ce_w, ab_w, sse_w, kld_w = loss_weights(self.nsamples, self.nlatent, self.alpha, self.beta)
        dt = self.dtype
        depths_in, tnf_in, ab_in, weights = (v.astype(dt) for v in (depths_in, tnf_in, ab_in, weights))
        ab_sse = ((ab_out - ab_in) ** 2).sum(axis=1)
        ce = -(np.log(depths_out + 1e-9) * depths_in).sum(axis=1)
        sse = ((tnf_out - tnf_in) ** 2).sum(axis=1)
        kld = 0.5 * (mu ** 2).sum(axis=1)
        # NOTE (reference quirk, encode.py:347): the per-row terms are [B] but ``weights`` is [B, 1], so
        # ``(...) * weights`` broadcasts to a [B, B] outer product and ``loss.mean()`` equals
        # mean_i(row_i) * mean_j(w_j): the contig weights only scale the batch loss by their mean.
        w = weights.reshape(-1)
        rows = (ce * ce_w + ab_sse * ab_w + sse * sse_w) + kld * kld_w
        loss_row = rows * w.mean()
        # data-parallel shard of a larger batch: normalise by the all-rank batch instead
        # (d loss / d rows_i = wsum_global / B_global^2; see vamb_amd/parallel.py)
        self._row_grad = None if global_wsum is None else float(global_wsum) / float(global_batch) ** 2
        self._loss_in = dict(depths_in=depths_in, depths_out=depths_out, tnf_in=tnf_in, tnf_out=tnf_out,
                             ab_in=ab_in, ab_out=ab_out, mu=mu, w=w)
        return (loss_row.mean(), (ab_sse * ab_w).mean(), (ce * ce_w).mean(), (sse * sse_w).mean(),
                (kld * kld_w).mean())

    # ---- backward -------------------------------------------------------------------------------
    def backward(self):
        """Gradients of ``loss.mean()`` w.r.t. every parameter (same names as the state_dict)."""
        L = self._loss_in
        ce_w, ab_w, sse_w, kld_w = loss_weights(self.nsamples, self.nlatent, self.alpha, self.beta)
        B = L["w"].shape[0]
        g = np.full((B, 1), L["w"].mean() / B, dtype=self.dtype)  # d mean(outer(rows, w)) / d rows_i
        if getattr(self, "_row_grad", None) is not None:
            g = np.full((B, 1), self._row_grad, dtype=self.dtype)
        S = self.nsamples
        p = L["depths_out"]
        dp = g * ce_w * (-L["depths_in"] / (p + 1e-9))
        dlogit = p * (dp - (p * dp).sum(axis=1, keepdims=True))
        dtnf = g * sse_w * 2.0 * (L["tnf_out"] - L["tnf_in"])
        dab = g * ab_w * 2.0 * (L["ab_out"] - L["ab_in"])
        drecon = np.concatenate([dlogit, dtnf, dab], axis=1)
        dmu_kld = g * kld_w * L["mu"]
        return self._backprop(drecon, dmu_kld)

    def _backprop(self, drecon, dmu_kld):
        """Chain rule from d loss / d reconstruction and the KLD part of d loss / d mu down to every parameter."""
        grads = {}
        st = self.state
        nl = len(self.nhiddens)
        tape = self._tape

        grads["outputlayer.weight"] = drecon.T @ tape["a_last"]
        grads["outputlayer.bias"] = drecon.sum(axis=0)
        da = drecon @ st["outputlayer.weight"]

        def hidden_bwd(da, t):
            norm, lin = t["norm"], t["lin"]
            grads[norm + ".weight"] = (da * t["xhat"]).sum(axis=0)
            grads[norm + ".bias"] = da.sum(axis=0)
            dxhat = da * st[norm + ".weight"]
            if self.bn_sync is not None:
                nc = dxhat.shape[1]
                tot = self.bn_sync(np.concatenate([dxhat.sum(axis=0), (dxhat * t["xhat"]).sum(axis=0)]))
                m1, m2 = tot[:nc] / t["n_glob"], tot[nc:] / t["n_glob"]
            else:
                m1, m2 = dxhat.mean(axis=0), (dxhat * t["xhat"]).mean(axis=0)
            dh = t["invstd"] * (dxhat - m1 - t["xhat"] * m2)
            if self.dropout > 0 and t["mask"] is not None:
                scale = self.dtype(1.0) / (self.dtype(1.0) - self.dtype(self.dropout))
                dr = dh * (t["mask"].astype(self.dtype) * scale)
            else:
                dr = dh
            dz = dr * np.where(t["z"] > 0, 1.0, LRELU_SLOPE)
            grads[lin + ".weight"] = dz.T @ t["a_prev"]
            grads[lin + ".bias"] = dz.sum(axis=0)
            return dz @ st[lin + ".weight"]

        for i in reversed(range(nl)):
            da = hidden_bwd(da, tape["hidden"][nl + i])
        dmu = da + dmu_kld
        grads["mu.weight"] = dmu.T @ tape["mu"]["a_prev"]
        grads["mu.bias"] = dmu.sum(axis=0)
        da = dmu @ st["mu.weight"]
        for i in reversed(range(nl)):
            da = hidden_bwd(da, tape["hidden"][i])
        self.grads = grads
        return grads

    # ---- optimizer ------------------------------------------------------------------------------
    def dadapt_step(self, grads=None, b1=0.9, b2=0.999, eps=1e-8, lr=1.0, growth_rate=float("inf")):
        """dadaptation.DAdaptAdam.step as Vamb configures it (see oracle/dadapt_restated.py)."""
        grads = self.grads if grads is None else grads
        dlr = self.d * lr
        sqrt_b2 = b2 ** 0.5
        sk_l1 = 0.0
        num_acum = 0.0
        for n in self.names:
            g = grads[n]
            o = self.opt[n]
            denom = np.sqrt(o["v"]) + eps
            num_acum += dlr * float((g * (o["s"] / denom)).sum())
            o["m"] = b1 * o["m"] + dlr * (1 - b1) * g
            o["v"] = b2 * o["v"] + (1 - b2) * g * g
            o["s"] = sqrt_b2 * o["s"] + dlr * (1 - sqrt_b2) * g
            sk_l1 += float(np.abs(o["s"]).sum())
        nw = sqrt_b2 * self.numerator_weighted + (1 - sqrt_b2) * num_acum
        if sk_l1 == 0:
            return
        d_hat = nw / ((1 - sqrt_b2) * sk_l1)
        self.d = max(self.d, min(d_hat, self.d * growth_rate))
        self.numerator_weighted = nw
        for n in self.names:
            o = self.opt[n]
            self.state[n] = self.state[n] - o["m"] / (np.sqrt(o["v"]) + eps)
        self.k += 1

    def train_step(self, depths, tnf, abundance, weights, eps, masks):
        do, to, ao, mu = self.forward(depths, tnf, abundance, eps=eps, masks=masks, train=True)
        losses = self.calc_loss(depths, do, tnf, to, abundance, ao, mu, weights)
        self.backward()
        self.dadapt_step()
        return losses

    # ---- encode ---------------------------------------------------------------------------------
    def encode(self, depths, tnf, abundance):
        """encode.py:442-484: eval-mode mu for every row, then clear the low 12 mantissa bits."""
        dt = self.dtype
        a = np.concatenate([depths, tnf, abundance], axis=1).astype(dt)
        for i in range(len(self.nhiddens)):
            a, _ = self._hidden_fwd(a, f"encoderlayers.{i}", f"encodernorms.{i}", None, False)
        mu = a @ self.state["mu.weight"].T + self.state["mu.bias"]
        lat = np.ascontiguousarray(mu.astype(np.float32))
        u = lat.view(np.uint32)
        u &= ~np.uint32(0xFFF)
        return lat
