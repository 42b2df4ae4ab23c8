"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the joint TaxVamb trainer (explicit backward).

Part of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

What is restated (the remainder after VAELabels / VAEConcat):
  * ``vamb.hloss_misc.Hierarchy.leaf_mask`` / ``ancestor_mask(strict=False)``            hloss_misc.py:20-124
  * ``vamb.hloss_misc.FlatSoftmaxNLL`` (the default ``hier_loss='flat_softmax'``)          hloss_misc.py:1102-1133
  * ``VAELabelsHLoss.calc_loss`` / ``VAEConcatHLoss.calc_loss``                            taxvamb_encode.py:348-355, 496-538
  * ``VAEVAEHLoss.calc_loss_joint`` + ``kld_gauss``                                        taxvamb_encode.py:541-548, 682-743
  * ``VAEVAE.trainepoch``'s step: seven passes through three networks, one backward()      semisupervised_encode.py:864-997
  * ``torch.optim.Adam(lr)`` over the three networks' parameters                           semisupervised_encode.py:1048-1053
Pinned by ``tests/golden/vaevae_*.npz`` -- recorded by running the REAL classes' own ``trainmodel`` under torch autograd with
injected dropout masks and noise (tests/golden/make_golden.py vaevae).

Facts of the reference that shape this file:
  * every ``logsigma`` is a zero tensor (semisupervised_encode.py:241, 507, 916), so ``kld_gauss(p, 0, q, 0)`` is
    ``0.5 * mean((p - q)^2)`` over all B x nlatent elements and VAELabels' KLD is ``0.5 * sum(mu^2, dim=1).mean()``;
  * the label logits the loss sees are the FIRST ``n_leaves`` output columns: the HLoss subclasses overwrite ``self.nlabels``
    with the number of leaves AFTER the layers were built ``max(n_nodes, 105)`` wide (taxvamb_encode.py:329, 477;
    ``_decode``'s ``narrow(1, .., self.nlabels)``, semisupervised_encode.py:236, 493) -- the other output columns never receive a
    gradient, and VAEConcatHLoss's narrow starts at ``nsamples + 104`` like its parent's;
  * ``(row terms [B] + scalars) * weights [B, 1]`` broadcasts to [B, B]: the mean is mean(rows + scalars) * mean(weights)
    (the same quirk as encode.py:347);
  * nobody calls ``.train()`` / ``.eval()`` in ``VAEVAE.trainepoch``: the three modules are in the mode their constructor left
    them in (training), so every one of the seven passes applies dropout, uses batch statistics and updates the running
    statistics -- also the passes whose outputs are thrown away (VAEJoint's decoder, the decoders of the two ``*_sup_s`` passes).
"""
from __future__ import annotations

import numpy as np

from vae_oracle import BN_EPS, LRELU_SLOPE, NTNF, OracleVAE, loss_weights, param_names  # noqa: F401


# ---- hierarchy ------------------------------------------------------------------------------------------------------
def leaf_mask(table_parent):
    """hloss_misc.py:51-58: nodes without children."""
    p = np.asarray(table_parent)
    n = len(p)
    nchild = np.zeros(n, dtype=int)
    for j in range(1, n):
        nchild[p[j]] += 1
    return nchild == 0


def leaf_masks_of_nodes(table_parent):
    """FlatSoftmaxNLL.__init__ (hloss_misc.py:1110-1113): [n_nodes][n_leaves] bool, True where the leaf is the node itself or
    one of its descendants."""
    p = np.asarray(table_parent)
    n = len(p)
    assert p[0] == -1 and np.all(p[1:] < np.arange(1, n)) and np.all(p[1:] >= 0)
    is_desc = np.zeros((n, n), dtype=bool)   # is_desc[j, i]: j is i or below i
    is_desc[0, 0] = True
    for j in range(1, n):
        is_desc[j] = is_desc[p[j]]
        is_desc[j, j] = True
    is_anc = is_desc.T
    return is_anc[:, leaf_mask(p)]


def flat_softmax_nll(scores, nodes, masks):
    """FlatSoftmaxNLL.forward with reduction='mean' (hloss_misc.py:1121-1133).
    Returns (mean loss, d mean loss / d scores)."""
    B = len(scores)
    m = scores.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(scores - m).sum(axis=1, keepdims=True))
    logp = scores - lse
    M = masks[nodes]
    neg = np.where(M, logp, -np.inf)
    mm = neg.max(axis=1, keepdims=True)
    logp_label = mm[:, 0] + np.log(np.exp(neg - mm).sum(axis=1))
    p = np.exp(logp)
    inside = np.where(M, p, 0.0)
    d = (p - inside / inside.sum(axis=1, keepdims=True)) / B
    return float((-logp_label).mean()), d


# ---- one network, pass by pass ---------------------------------------------------------------------------------------
class _Net:
    """One of the three modules: parameters + running statistics (``state``), Adam moments, and per-pass tapes."""

    def __init__(self, nsamples, width, nhiddens, nlatent, dropout, state, dtype):
        # OracleVAE supplies _hidden_fwd (Linear -> LeakyReLU -> dropout -> BatchNorm with torch's train / eval semantics)
        self.eng = OracleVAE(max(nsamples, 1), nhiddens, nlatent, alpha=0.5, beta=1.0, dropout=dropout, state=state, dtype=dtype)
        self.state = self.eng.state
        self.nl = len(nhiddens)
        self.width = width
        self.names = param_names(nhiddens)
        self.opt = {n: dict(m=np.zeros_like(self.state[n]), v=np.zeros_like(self.state[n])) for n in self.names}
        self.grads = {n: np.zeros_like(self.state[n]) for n in self.names}
        self.dtype = dtype
        self.dropout = dropout

    def zero_grad(self):
        for n in self.names:
            self.grads[n] = np.zeros_like(self.state[n])

    def encode(self, x, masks, train=True):
        a = x.astype(self.dtype)
        tape = []
        for i in range(self.nl):
            a, t = self.eng._hidden_fwd(a, f"encoderlayers.{i}", f"encodernorms.{i}", None if masks is None else masks[i], train)
            tape.append(t)
        mu = a @ self.state["mu.weight"].T + self.state["mu.bias"]
        return mu, dict(hidden=tape, a_prev=a)

    def decode(self, z, masks, train=True):
        a = z
        tape = []
        for i in range(self.nl):
            a, t = self.eng._hidden_fwd(a, f"decoderlayers.{i}", f"decodernorms.{i}", None if masks is None else masks[i], train)
            tape.append(t)
        recon = a @ self.state["outputlayer.weight"].T + self.state["outputlayer.bias"]
        return recon, dict(hidden=tape, a_last=a)

    def _hidden_bwd(self, da, t):
        st, g = self.state, self.grads
        norm, lin = t["norm"], t["lin"]
        g[norm + ".weight"] = g[norm + ".weight"] + (da * t["xhat"]).sum(axis=0)
        g[norm + ".bias"] = g[norm + ".bias"] + da.sum(axis=0)
        dxhat = da * st[norm + ".weight"]
        m1, m2 = dxhat.mean(axis=0), (dxhat * t["xhat"]).mean(axis=0)
        dh = t["invstd"] * (dxhat - m1 - t["xhat"] * m2)
        if self.dropout > 0 and t["mask"] is not None:
            scale = self.dtype(1.0) / (self.dtype(1.0) - self.dtype(self.dropout))
            dh = dh * (t["mask"].astype(self.dtype) * scale)
        dz = dh * np.where(t["z"] > 0, 1.0, LRELU_SLOPE)
        g[lin + ".weight"] = g[lin + ".weight"] + dz.T @ t["a_prev"]
        g[lin + ".bias"] = g[lin + ".bias"] + dz.sum(axis=0)
        return dz @ st[lin + ".weight"]

    def decode_bwd(self, tape, drecon):
        """Accumulates the decoder-side gradients; returns d loss / d z."""
        g = self.grads
        g["outputlayer.weight"] = g["outputlayer.weight"] + drecon.T @ tape["a_last"]
        g["outputlayer.bias"] = g["outputlayer.bias"] + drecon.sum(axis=0)
        da = drecon @ self.state["outputlayer.weight"]
        for t in reversed(tape["hidden"]):
            da = self._hidden_bwd(da, t)
        return da

    def encode_bwd(self, tape, dmu):
        g = self.grads
        g["mu.weight"] = g["mu.weight"] + dmu.T @ tape["a_prev"]
        g["mu.bias"] = g["mu.bias"] + dmu.sum(axis=0)
        da = dmu @ self.state["mu.weight"]
        for t in reversed(tape["hidden"]):
            da = self._hidden_bwd(da, t)

    def adam_step(self, t, lr, b1=0.9, b2=0.999, eps=1e-8):
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        for n in self.names:
            g, o = self.grads[n], self.opt[n]
            o["m"] = o["m"] + (g - o["m"]) * (1 - b1)
            o["v"] = b2 * o["v"] + (1 - b2) * g * g
            self.state[n] = self.state[n] - (lr / bc1) * (o["m"] / (np.sqrt(o["v"]) / np.sqrt(bc2) + eps))


METRICS = ["loss_vamb", "ab_vamb", "ce_vamb", "sse_vamb", "kld_vamb", "loss_labels", "ce_labels_labels", "kld_labels",
           "correct_labels_labels", "loss_joint", "ce_joint", "sse_joint", "ce_labels_joint", "kld_vamb_joint", "kld_labels_joint",
           "correct_labels_joint", "loss"]   # semisupervised_encode.py:830-848


class OracleVAEVAE:
    """VAEVAEHLoss(nsamples, nlabels=len(nodes), nodes, table_parent, ...) -- taxvamb_encode.py:579-628.
    ``table_parent`` an int: the one-hot base class VAEVAE(nsamples, nlabels) as its code reads (semisupervised_encode.py:721-827;
    the reference itself cannot run it, see vamb_amd/semisupervised_encode.py:VAEVAE) -- CrossEntropyLoss over all label columns,
    i.e. the flat-softmax loss with the identity as mask -- plus the ``correct_labels`` counts."""

    def __init__(self, nsamples, table_parent, nhiddens, nlatent, alpha, beta, dropout, states, dtype=np.float64):
        if alpha is None:
            alpha = 0.15 if nsamples > 1 else 0.50
        self.nsamples, self.nlatent, self.alpha, self.beta = nsamples, nlatent, alpha, beta
        self.one_hot = isinstance(table_parent, (int, np.integer))
        if self.one_hot:
            self.n_nodes = int(table_parent)
            self.NL = max(self.n_nodes, 105)
            self.masks = np.eye(self.NL, dtype=bool)
        else:
            self.n_nodes = len(table_parent)
            self.NL = max(self.n_nodes, 105)                 # width of the label block (taxvamb_encode.py:593)
            self.masks = leaf_masks_of_nodes(table_parent)
        self.n_leaves = self.masks.shape[1]
        self.dtype = dtype
        S = nsamples
        # (hidden widths may differ between the networks -- with nhiddens=None the reference's VAELabels over <= 105 label columns
        # takes the single-sample default [256, 256] beside [512, 512] -- the depth may not: a dict gives them per network)
        nh = nhiddens if isinstance(nhiddens, dict) else dict(VAEVamb=nhiddens, VAELabels=nhiddens, VAEJoint=nhiddens)
        self.vamb = _Net(S, S + NTNF + 1, nh["VAEVamb"], nlatent, dropout, states["VAEVamb"], dtype)
        self.labels = _Net(0, self.NL, nh["VAELabels"], nlatent, dropout, states["VAELabels"], dtype)
        self.joint = _Net(S, S + NTNF + 1 + self.NL, nh["VAEJoint"], nlatent, dropout, states["VAEJoint"], dtype)
        self.t = 0

    def _onehot(self, nodes):
        return np.eye(self.NL, dtype=self.dtype)[nodes]

    def _vamb_outputs(self, recon):
        """VAE._decode (encode.py:288-304): softmax over the depth columns -- ALSO for a single sample (the output is then the
        constant 1 and the cross-entropy -log(1 + 1e-9) * x: zero in float32); only VAEConcat._decode skips it for nsamples == 1
        (semisupervised_encode.py:497-500), and no loss is ever computed on that decoder here."""
        S = self.nsamples
        out = recon.copy()
        e = np.exp(recon[:, :S] - recon[:, :S].max(axis=1, keepdims=True))
        out[:, :S] = e / e.sum(axis=1, keepdims=True)
        return out

    def _vamb_recon_terms(self, out, x, g):
        """Row terms of VAE.calc_loss without the KLD, and d(sum_i g * row_i) / d reconstruction."""
        S = self.nsamples
        ce_w, ab_w, sse_w, _ = loss_weights(S, self.nlatent, self.alpha, self.beta)
        p, d_in = out[:, :S], x[:, :S]
        t_out, t_in = out[:, S:S + NTNF], x[:, S:S + NTNF]
        a_out, a_in = out[:, S + NTNF:S + NTNF + 1], x[:, S + NTNF:S + NTNF + 1]
        ce = -(np.log(p + 1e-9) * d_in).sum(axis=1)
        sse = ((t_out - t_in) ** 2).sum(axis=1)
        ab = ((a_out - a_in) ** 2).sum(axis=1)
        drecon = np.zeros_like(out)
        dp = g * ce_w * (-d_in / (p + 1e-9))
        drecon[:, :S] = p * (dp - (p * dp).sum(axis=1, keepdims=True))
        drecon[:, S:S + NTNF] = g * sse_w * 2.0 * (t_out - t_in)
        drecon[:, S + NTNF:S + NTNF + 1] = g * ab_w * 2.0 * (a_out - a_in)
        return ce, sse, ab, (ce_w, ab_w, sse_w), drecon

    def train_step(self, unsup, unsup_nodes, sup, sup_nodes, rnd, lr=1e-3):
        """One batch of VAEVAE.trainepoch.
        unsup / sup: dicts with depths, tnf, abundance, weights ([B, .] float arrays); *_nodes: [B] node indices.
        rnd: per pass (joint, vamb_x, labels_x, vamb_u, vamb_s, labels_u, labels_s) a dict(masks=[...], eps=[B, L]);
             the ``_x`` passes only decode (nl masks), the others hold 2 * nl masks (encoder then decoder layers).
        Returns the 17 metrics in METRICS order."""
        dt, S, L = self.dtype, self.nsamples, self.nlatent
        nl = self.vamb.nl
        kld_w = 1 / (L * self.beta)
        xu = np.concatenate([unsup["depths"], unsup["tnf"], unsup["abundance"]], axis=1).astype(dt)
        xs = np.concatenate([sup["depths"], sup["tnf"], sup["abundance"]], axis=1).astype(dt)
        wu, ws = unsup["weights"].astype(dt).reshape(-1), sup["weights"].astype(dt).reshape(-1)
        B = len(xs)
        oh_u, oh_s = self._onehot(unsup_nodes), self._onehot(sup_nodes)
        for net in (self.vamb, self.labels, self.joint):
            net.zero_grad()

        def split(r):
            m = r["masks"]
            return (None, None) if m is None else (m[:nl], m[nl:])

        # -- the seven passes, in the reference's order (:899-928)
        je, jd = split(rnd["joint"])
        mu_sup, tape_j = self.joint.encode(np.concatenate([xs, oh_s], axis=1), je)
        self.joint.decode(mu_sup + rnd["joint"]["eps"].astype(dt), jd)             # outputs unused, statistics updated
        rec_vx, tape_vx = self.vamb.decode(mu_sup + rnd["vamb_x"]["eps"].astype(dt), rnd["vamb_x"]["masks"])
        rec_lx, tape_lx = self.labels.decode(mu_sup + rnd["labels_x"]["eps"].astype(dt), rnd["labels_x"]["masks"])
        ue, ud = split(rnd["vamb_u"])
        mu_vu, tape_vu_e = self.vamb.encode(xu, ue)
        rec_vu, tape_vu_d = self.vamb.decode(mu_vu + rnd["vamb_u"]["eps"].astype(dt), ud)
        se, sd = split(rnd["vamb_s"])
        mu_vs, tape_vs_e = self.vamb.encode(xs, se)
        self.vamb.decode(mu_vs + rnd["vamb_s"]["eps"].astype(dt), sd)
        ue, ud = split(rnd["labels_u"])
        mu_lu, tape_lu_e = self.labels.encode(oh_u, ue)
        rec_lu, tape_lu_d = self.labels.decode(mu_lu + rnd["labels_u"]["eps"].astype(dt), ud)
        se, sd = split(rnd["labels_s"])
        mu_ls, tape_ls_e = self.labels.encode(oh_s, se)
        self.labels.decode(mu_ls + rnd["labels_s"]["eps"].astype(dt), sd)

        # -- VAEVamb.calc_loss on the unsupervised batch (encode.py:316-357)
        gu = wu.mean() / B
        out_vu = self._vamb_outputs(rec_vu)
        ce, sse, ab, (ce_w, ab_w, sse_w), drec = self._vamb_recon_terms(out_vu, xu, gu)
        kld = 0.5 * (mu_vu ** 2).sum(axis=1)
        m = dict(ab_vamb=(ab * ab_w).mean(), ce_vamb=(ce * ce_w).mean(), sse_vamb=(sse * sse_w).mean(), kld_vamb=(kld * kld_w).mean())
        m["loss_vamb"] = (((ce * ce_w + ab * ab_w + sse * sse_w) + kld * kld_w) * wu.mean()).mean()
        dz = self.vamb.decode_bwd(tape_vu_d, drec)
        self.vamb.encode_bwd(tape_vu_e, dz + gu * kld_w * mu_vu)

        # -- VAELabelsHLoss.calc_loss on the unsupervised labels (taxvamb_encode.py:348-355)
        cel, dsc = flat_softmax_nll(rec_lu[:, :self.n_leaves], unsup_nodes, self.masks)
        kld_l = 0.5 * (mu_lu ** 2).sum(axis=1).mean()
        m.update(ce_labels_labels=cel, kld_labels=kld_l, loss_labels=cel + kld_l * kld_w,
                 correct_labels_labels=float((rec_lu[:, :self.n_leaves].argmax(axis=1) == unsup_nodes).sum()) if self.one_hot else 0.0)
        drec = np.zeros_like(rec_lu)
        drec[:, :self.n_leaves] = dsc
        dz = self.labels.decode_bwd(tape_lu_d, drec)
        self.labels.encode_bwd(tape_lu_e, dz + kld_w * mu_lu / B)

        # -- VAEVAEHLoss.calc_loss_joint on the supervised batch (taxvamb_encode.py:682-743)
        wm = ws.mean()
        out_vx = self._vamb_outputs(rec_vx)
        ce, sse, ab, _, drec_v = self._vamb_recon_terms(out_vx, xs, wm / B)
        cel_j, dsc = flat_softmax_nll(rec_lx[:, :self.n_leaves], sup_nodes, self.masks)
        kld_v = 0.5 * ((mu_sup - mu_vs) ** 2).mean()
        kld_lb = 0.5 * ((mu_sup - mu_ls) ** 2).mean()
        recon_rows = ((ce * ce_w + ab * ab_w) + sse * sse_w) + cel_j
        m.update(loss_joint=((recon_rows + (kld_v + kld_lb) * kld_w) * wm).mean(), ce_joint=ce.mean(), sse_joint=sse.mean(),
                 ce_labels_joint=cel_j, kld_vamb_joint=kld_v, kld_labels_joint=kld_lb,
                 correct_labels_joint=float((rec_lx[:, :self.n_leaves].argmax(axis=1) == sup_nodes).sum()) if self.one_hot else 0.0)
        drec_l = np.zeros_like(rec_lx)
        drec_l[:, :self.n_leaves] = dsc * wm
        dmu = self.vamb.decode_bwd(tape_vx, drec_v) + self.labels.decode_bwd(tape_lx, drec_l)
        gk = wm * kld_w / (B * L)
        self.joint.encode_bwd(tape_j, dmu + gk * ((mu_sup - mu_vs) + (mu_sup - mu_ls)))
        self.vamb.encode_bwd(tape_vs_e, -gk * (mu_sup - mu_vs))
        self.labels.encode_bwd(tape_ls_e, -gk * (mu_sup - mu_ls))
        m["loss"] = m["loss_joint"] + m["loss_vamb"] + m["loss_labels"]

        self.t += 1
        for net in (self.vamb, self.labels, self.joint):
            net.adam_step(self.t, lr)
        self.mu_sup = mu_sup
        return [float(m[k]) for k in METRICS]

    def encode_joint(self, depths, tnf, abundance, nodes):
        """VAEConcat.encode (semisupervised_encode.py:651-697): eval-mode mu of VAEJoint, low 12 mantissa bits cleared."""
        x = np.concatenate([depths, tnf, abundance, self._onehot(nodes)], axis=1)
        mu, _ = self.joint.encode(x, None, train=False)
        lat = np.ascontiguousarray(mu.astype(np.float32))
        u = lat.view(np.uint32)
        u &= ~np.uint32(0xFFF)
        return lat
