#!/usr/bin/env python
"""Generate the golden vectors in this directory by running the REAL reference code.

Build-container only: needs /root/reference (read-only) and loads
``vamb/{vambtools,cluster,encode}.py`` through ``oracle/ref_harness.py`` (stubs only for loguru,
vambcore.overwrite_matrix and dadaptation.DAdaptAdam -- the latter restated, PARITY UNPINNED).
The GPU box never runs this; tests read the committed .npz files.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py cluster    # only one family (cluster | cluster_large | prep | vae | semisup | vaevae | tnf | e2e)

Environment recorded in golden_manifest.json (torch / numpy versions, thread count).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import fixture_defs as fd  # noqa: E402
import ref_harness  # noqa: E402
import vae_oracle  # noqa: E402


def gen_cluster(cl, cases=None):
    out = {}
    for name in (fd.CLUSTER_CASES if cases is None else cases):
        mat, lens, kw = fd.cluster_inputs(name)
        clusters = list(cl.ClusterGenerator(mat.copy(), lens, **kw))
        packed = fd.pack_stream(clusters)
        # np.argsort is unstable (cluster.py:275): record the seed order this CPU produced
        packed["order_sha256"] = np.array(hashlib.sha256(
            np.argsort(lens)[::-1].astype(np.int64).tobytes()).hexdigest())
        np.savez_compressed(os.path.join(HERE, f"cluster_{name}.npz"), **packed)
        kinds = np.bincount(packed["kind"], minlength=3)
        out[name] = dict(n_clusters=len(clusters), normal=int(kinds[0]), loner=int(kinds[1]),
                         fallback=int(kinds[2]))
        print("cluster", name, out[name])
    return out


def gen_tnf():
    """kernel.npz and Composition._project from the REAL reference (vamb/parsecontigs.py executed unmodified); the counts by
    the definition the reference's own test pins vambcore.kmercounts to (test/test_vambtools.py:137-151) -- the Rust wheel is
    not in this image."""
    import importlib.util
    import itertools

    ref_harness.load_reference()
    spec = importlib.util.spec_from_file_location(
        "vamb.parsecontigs", os.path.join(ref_harness.REFERENCE_ROOT, "vamb", "parsecontigs.py"))
    pc = importlib.util.module_from_spec(spec)
    sys.modules["vamb.parsecontigs"] = pc
    spec.loader.exec_module(pc)
    vt = sys.modules["vamb.vambtools"]
    indexof = {"".join(ncs): idx for (idx, ncs) in enumerate(itertools.product("ACGT", repeat=4))}
    seqs = fd.tnf_sequences()
    counts = np.zeros((len(seqs), 256), dtype=np.uint32)
    for r, seq in enumerate(seqs):
        text = seq.decode()
        for i in range(len(text) - 3):
            ind = indexof.get(text[i:i + 4].upper())
            if ind is not None:
                counts[r, ind] += 1
    proj = pc.Composition._project(counts.astype(np.float32))
    tnf = np.ascontiguousarray(proj, dtype=np.float32).copy()
    flat = tnf.reshape(-1)           # a view: mask_lower_bits works in place (vambtools.py:324-330)
    vt.mask_lower_bits(flat, 12)
    np.savez_compressed(os.path.join(HERE, "tnf_case.npz"), kernel=np.asarray(pc._KERNEL, dtype=np.float32), counts=counts,
                        projected=np.asarray(proj, dtype=np.float32), tnf=tnf)
    print("tnf", counts.shape, proj.shape)
    return dict(n=len(seqs))


def gen_prep(en):
    out = {}
    for name in fd.PREP_CASES:
        ab, tnf, lens = fd.prep_inputs(name)
        dl = en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=16)
        d, t, a, w = (x.numpy().copy() for x in dl.dataset.tensors)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), depths=d, tnf=t, total_abundance=a, weights=w)
        out[name] = dict(n=len(d))
        print("prep", name, d.shape)
    return out


def gen_vae(en, only=None):
    import torch
    import dadaptation  # the stub installed by ref_harness (oracle/dadapt_restated.py)

    out = {}
    for name, c in fd.VAE_CASES.items():
        if only is not None and name not in only:
            continue
        ab, tnf, lens = fd.vae_inputs(name)
        dl = en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=c["batch"])
        depths, tnfz, totab, weights = dl.dataset.tensors
        B = c["batch"]
        vae = en.VAE(c["nsamples"], nhiddens=list(c["nhiddens"]), nlatent=c["nlatent"], alpha=c["alpha"],
                     beta=c["beta"], dropout=c["dropout"], seed=0)
        st0 = vae_oracle.init_state(c["nsamples"], c["nhiddens"], c["nlatent"], c["seed"])
        vae.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st0.items()})
        masks, eps = fd.vae_randomness(name)
        mask_q, eps_q = [], []

        class InjectedDropout(torch.nn.Module):
            def forward(self, x):
                if not self.training or c["dropout"] == 0:
                    return x
                m = mask_q.pop(0)
                scale = np.float32(1.0) / (np.float32(1.0) - np.float32(c["dropout"]))
                return x * torch.from_numpy(m.astype(np.float32) * scale)

        vae.dropoutlayer = InjectedDropout()
        vae.reparameterize = lambda mu: (mu + torch.from_numpy(eps_q.pop(0))) if eps_q else mu
        opt = dadaptation.DAdaptAdam(vae.parameters(), decouple=True)
        keep = c.get("rows_keep")

        def rows_of(x):   # per-row outputs of a big case: the first rows + the norm of everything
            x = np.asarray(x)
            return x.copy() if keep is None else x[:keep].copy()

        if c.get("store_inputs", True):
            rec = dict(depths=depths.numpy().copy(), tnf=tnfz.numpy().copy(), total_abundance=totab.numpy().copy(),
                       weights=weights.numpy().copy())
        else:   # checksums of the reference-normalised inputs (exact fp64 sums of the float32 values and of their squares)
            rec = {}
            for k, v in (("depths", depths), ("tnf", tnfz), ("total_abundance", totab), ("weights", weights)):
                v64 = v.numpy().astype(np.float64)
                rec["input_sum/" + k] = np.array(v64.sum())
                rec["input_sumsq/" + k] = np.array((v64 * v64).sum())
                rec["input_head/" + k] = v.numpy()[:8].copy()
        losses, ds = [], []
        vae.train()
        for step in range(c["steps"]):
            mask_q[:] = list(masks[step])
            eps_q[:] = [eps[step]]
            d_in, t_in, a_in, w_in = depths[:B], tnfz[:B], totab[:B], weights[:B]
            opt.zero_grad()
            do, to, ao, mu = vae(d_in, t_in, a_in)
            ls = vae.calc_loss(d_in, do, t_in, to, a_in, ao, mu, w_in)
            ls[0].backward()
            if step == 0:
                for key, val in (("step0_depths_out", do), ("step0_tnf_out", to), ("step0_ab_out", ao), ("step0_mu", mu)):
                    val = val.detach().numpy()
                    rec[key] = rows_of(val)
                    if keep is not None:
                        rec[key + "_norm"] = np.array(np.sqrt((val.astype(np.float64) ** 2).sum()))
                for pname, p in vae.named_parameters():
                    g = p.grad.detach().numpy()
                    if c["store"] == "full":
                        rec["grad0/" + pname] = g.copy()
                    else:
                        rec["grad0_norm/" + pname] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
                        rec["grad0_head/" + pname] = g.reshape(-1)[:64].copy()
            opt.step()
            losses.append([float(x.item()) for x in ls])
            ds.append(opt.param_groups[0]["d"])
        rec["losses"] = np.array(losses, np.float64)
        rec["d_after"] = np.array(ds, np.float64)
        rec["numerator_weighted"] = np.array(opt.param_groups[0]["numerator_weighted"], np.float64)
        for k, v in vae.state_dict().items():
            v = v.numpy()
            if c["store"] == "full" or v.size <= 4096:
                rec["final/" + k] = v.copy()
            else:
                rec["final_norm/" + k] = np.array(np.sqrt((v.astype(np.float64) ** 2).sum()))
                rec["final_head/" + k] = v.reshape(-1)[:64].copy()
        vae.eval()
        lat = vae.encode(dl)
        rec["latent"] = rows_of(lat)
        if keep is not None:
            rec["latent_norm"] = np.array(np.sqrt((lat.astype(np.float64) ** 2).sum()))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        out[name] = dict(loss0=losses[0][0], loss_last=losses[-1][0], d_last=ds[-1])
        print("vae", name, out[name])
    return out


def gen_semisup():
    """VAEConcat / VAELabels of the real reference (semisupervised_encode.py) with injected dropout masks and noise."""
    import torch
    import dadaptation  # the stub installed by ref_harness (oracle/dadapt_restated.py)

    ss = ref_harness.load_reference_module("semisupervised_encode")
    out = {}
    for name, c in fd.SEMISUP_CASES.items():
        ab, tnf, lens, labels = fd.semisup_inputs(name)
        B, NL = c["batch"], fd.semisup_width(name)
        masks, eps = fd.semisup_randomness(name)
        mask_q, eps_q = [], []

        class InjectedDropout(torch.nn.Module):
            def forward(self, x):
                if not self.training or c["dropout"] == 0:
                    return x
                m = mask_q.pop(0)
                scale = np.float32(1.0) / (np.float32(1.0) - np.float32(c["dropout"]))
                return x * torch.from_numpy(m.astype(np.float32) * scale)

        rec = {}
        if c["kind"] == "concat":
            dl = ss.make_dataloader_concat(ab.copy(), tnf.copy(), lens, labels, batchsize=B)
            depths, tnfz, totab, weights, lab_int = dl.dataset.tensors
            vae = ss.VAEConcat(c["nsamples"], NL, nhiddens=list(c["nhiddens"]), nlatent=c["nlatent"], alpha=c["alpha"],
                               beta=c["beta"], dropout=c["dropout"])
            width = c["nsamples"] + 104 + NL
            rec.update(depths=depths.numpy().copy(), tnf=tnfz.numpy().copy(), total_abundance=totab.numpy().copy(),
                       weights=weights.numpy().copy())
            opt = dadaptation.DAdaptAdam(vae.parameters(), decouple=True)
        else:
            dl = ss.make_dataloader_labels(ab.copy(), tnf.copy(), lens, labels, batchsize=B)
            (lab_int,) = dl.dataset.tensors
            vae = ss.VAELabels(NL, nhiddens=list(c["nhiddens"]), nlatent=c["nlatent"], alpha=c["alpha"], beta=c["beta"],
                               dropout=c["dropout"])
            width = NL
            opt = torch.optim.Adam(vae.parameters(), lr=c["lrate"])
        rec["labels"] = lab_int.numpy().astype(np.int64)
        rec["alpha"] = np.array(vae.alpha)
        st0 = vae_oracle.init_state(0, c["nhiddens"], c["nlatent"], c["seed"], width=width)
        vae.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st0.items()})
        vae.dropoutlayer = InjectedDropout()
        vae.reparameterize = lambda mu: (mu + torch.from_numpy(eps_q.pop(0))) if eps_q else mu
        onehot = torch.nn.functional.one_hot(lab_int[:B], num_classes=NL).float()
        losses = []
        vae.train()
        for step in range(c["steps"]):
            mask_q[:] = list(masks[step])
            eps_q[:] = [eps[step]]
            opt.zero_grad()
            if c["kind"] == "concat":
                d_in, t_in, a_in, w_in = depths[:B], tnfz[:B], totab[:B], weights[:B]
                do, to, ao, lo, mu, logsigma = vae(d_in, t_in, a_in, onehot)
                loss, ce, sse, cel, kld, correct = vae.calc_loss(d_in, do, t_in, to, a_in, ao, onehot, lo, mu, logsigma, w_in)
                loss.mean().backward()
                losses.append([loss.mean().item(), ce.mean().item(), sse.mean().item(), cel.item(), kld.mean().item(),
                               float(correct.item())])
                outs = dict(depths_out=do, tnf_out=to, ab_out=ao, labels_out=lo, mu=mu)
            else:
                lo, mu, logsigma = vae(onehot)
                loss, cel, kld, correct = vae.calc_loss(onehot, lo, mu, logsigma)
                loss.backward()
                losses.append([loss.item(), 0.0, 0.0, cel.item(), kld.item(), float(correct.item())])
                outs = dict(labels_out=lo, mu=mu)
            if step == 0:
                for k, v in outs.items():
                    rec["step0_" + k] = v.detach().numpy().copy()
                for pname, p in vae.named_parameters():
                    rec["grad0/" + pname] = p.grad.detach().numpy().copy()
            opt.step()
        rec["losses"] = np.array(losses, np.float64)   # loss, ce (raw mean), sse (raw mean), ce_labels, kld (raw mean), correct
        for k, v in vae.state_dict().items():
            rec["final/" + k] = v.numpy().copy()
        vae.eval()
        rec["latent"] = vae.encode(dl)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        out[name] = dict(loss0=losses[0][0], loss_last=losses[-1][0], correct_last=losses[-1][5])
        print("semisup", name, out[name])
    return out


def gen_vaevae(en):
    """VAEVAEHLoss of the real reference (taxvamb_encode.py:551-743 on semisupervised_encode.py:700-1084): its own loaders and its
    own ``trainmodel`` (one epoch = `steps` optimiser steps), with injected dropout masks / noise per network and pass; the
    per-step metrics are the return values of the three loss functions, the step-0 gradients are read off the parameters inside
    the first ``Adam.step``."""
    import torch

    ss = ref_harness.load_reference_module("semisupervised_encode")
    tx = ref_harness.load_reference_module("taxvamb_encode")
    out = {}
    for name, c in fd.VAEVAE_CASES.items():
        ab, tnf, lens, nodes, parents = fd.vaevae_inputs(name)
        B, S, N = c["batch"], c["nsamples"], len(parents)
        NL = max(N, 105)
        node_names = [f"n{i}" for i in range(N)]
        dl_v = en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B)
        jr = c.get("joint_rows", c["n"])   # < n: the joint loader holds other rows than the vamb / labels loaders
        dl_j = tx.make_dataloader_concat_hloss(ab[:jr].copy(), tnf[:jr].copy(), lens[:jr], nodes[:jr], N, parents, batchsize=B)
        dl_l = tx.make_dataloader_labels_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B)
        dl = tx.make_dataloader_semisupervised_hloss(dl_j, dl_v, dl_l, N, parents, (S, 103, 1, N), c["perm_seed"], batchsize=B)
        assert len(dl) == c["steps"], (len(dl), c["steps"])
        vae = tx.VAEVAEHLoss(S, N, node_names, parents, nhiddens=list(c["nhiddens"]), nlatent=c["nlatent"], alpha=c["alpha"],
                             beta=c["beta"], dropout=c["dropout"])
        assert vae.VAELabels.nlabels == vae.VAEJoint.nlabels == int(np.sum(~np.isin(np.arange(N), parents)))
        widths = dict(VAEVamb=None, VAELabels=NL, VAEJoint=S + 104 + NL)
        nets = dict(VAEVamb=vae.VAEVamb, VAELabels=vae.VAELabels, VAEJoint=vae.VAEJoint)
        for i, (k, net) in enumerate(nets.items()):
            st0 = vae_oracle.init_state(S if k != "VAELabels" else 0, c["nhiddens"], c["nlatent"], c["seed"] + i, width=widths[k])
            net.load_state_dict({kk: torch.from_numpy(np.array(v)) for kk, v in st0.items()})
        rnd = fd.vaevae_randomness(name)
        order = dict(VAEVamb=("vamb_x", "vamb_u", "vamb_s"), VAELabels=("labels_x", "labels_u", "labels_s"), VAEJoint=("joint",))
        queues = {}
        for k, net in nets.items():
            mq = [m for step in rnd for p in order[k] for m in step[p]["masks"]]
            eq = [step[p]["eps"] for step in rnd for p in order[k]]
            queues[k] = (mq, eq)

            class InjectedDropout(torch.nn.Module):
                def __init__(self, q):
                    super().__init__()
                    self.q = q

                def forward(self, x):
                    if not self.training or c["dropout"] == 0:
                        return x
                    m = self.q.pop(0)
                    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(c["dropout"]))
                    return x * torch.from_numpy(m.astype(np.float32) * scale)

            net.dropoutlayer = InjectedDropout(mq)
            net.reparameterize = (lambda q: (lambda mu: (mu + torch.from_numpy(q.pop(0))) if q else mu))(eq)
        rec = {}
        names10 = ("unsup_depths", "unsup_tnf", "unsup_abundance", "unsup_weights", "unsup_nodes", "sup_depths", "sup_tnf",
                   "sup_abundance", "sup_weights", "sup_nodes")
        for k, t in zip(names10, dl.dataset.tensors):
            rec[k] = t.numpy().copy()
        rec["parents"] = np.array(parents, np.int64)
        rec["alpha"] = np.array(vae.VAEVamb.alpha)
        steps_rec = []
        cur = {}

        def wrap(fn, key, grab=None):
            def inner(*a, **kw):
                r = fn(*a, **kw)
                cur[key] = [float(x.item()) for x in r]
                if grab is not None and "step0_mu_sup" not in rec:
                    grab(a)
                if key == "joint":
                    v, lab, j = cur["vamb"], cur["labels"], cur["joint"]
                    steps_rec.append(v + lab + j + [j[0] + v[0] + lab[0]])
                return r
            return inner

        def grab_joint(a):
            rec["step0_depths_out_x"] = a[1].detach().numpy().copy()
            rec["step0_tnf_out_x"] = a[3].detach().numpy().copy()
            rec["step0_labels_out_x"] = a[7].detach().numpy().copy()
            rec["step0_mu_sup"] = a[8].detach().numpy().copy()
            rec["step0_mu_vamb_sup"] = a[10].detach().numpy().copy()
            rec["step0_mu_labels_sup"] = a[12].detach().numpy().copy()

        vae.VAEVamb.calc_loss = wrap(vae.VAEVamb.calc_loss, "vamb")
        vae.VAELabels.calc_loss = wrap(vae.VAELabels.calc_loss, "labels")
        vae.calc_loss_joint = wrap(vae.calc_loss_joint, "joint", grab_joint)
        real_adam = ss._Adam

        class RecordingAdam(real_adam):
            def step(self, *a, **kw):
                if "grad0/VAEVamb/mu.weight" not in rec:
                    for k, net in nets.items():
                        for pname, p in net.named_parameters():
                            rec[f"grad0/{k}/{pname}"] = (torch.zeros_like(p) if p.grad is None else p.grad).detach().numpy().copy()
                return super().step(*a, **kw)

        ss._Adam = RecordingAdam
        try:
            vae.trainmodel(dl, nepochs=1, lrate=c["lrate"], batchsteps=None)
        finally:
            ss._Adam = real_adam
        assert len(steps_rec) == c["steps"] and not any(q[0] or q[1] for q in queues.values())
        rec["losses"] = np.array(steps_rec, np.float64)   # [steps][17] in the order of trainepoch's metrics list (:830-848)
        for k, net in nets.items():
            for kk, v in net.state_dict().items():
                rec[f"final/{k}/{kk}"] = v.numpy().copy()
        for net in nets.values():
            net.eval()
        rec["latent_joint"] = vae.VAEJoint.encode(dl_j)
        rec["latent_vamb"] = vae.VAEVamb.encode(dl_v)
        rec["joint_depths"], rec["joint_tnf"], rec["joint_abundance"], rec["joint_weights"], rec["joint_nodes"] = (
            t.numpy().copy() for t in dl_j.dataset.tensors)
        if jr != c["n"]:   # the rows latent_vamb was encoded from (otherwise they are the joint loader's)
            rec["vamb_depths"], rec["vamb_tnf"], rec["vamb_abundance"], rec["vamb_weights"] = (
                t.numpy().copy() for t in dl_v.dataset.tensors)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        out[name] = dict(loss0=steps_rec[0][-1], loss_last=steps_rec[-1][-1], n_leaves=int(vae.VAELabels.nlabels))
        print("vaevae", name, out[name])
    return out


def gen_e2e():
    """End-to-end runs of the real reference over several model seeds: loss curves + bin quality.  Free-running
    RNG, 8 threads (the CLI default, vamb/__main__.py:27-28): the stored numbers are a SPREAD to land in, not values to match."""
    import e2e_reference as e2e

    out = {}
    for name, c in fd.E2E_CASES.items():
        runs = [e2e.run_reference(c["n"], c["nsamples"], c["nepochs"], c["batchsize"], c["batchsteps"], seed,
                                  c["data_seed"], threads=8) for seed in c["model_seeds"]]
        rec = dict(losses=np.stack([r["losses"] for r in runs]), model_seeds=np.array(c["model_seeds"], np.int64))
        for k in runs[0]:
            if k != "losses":
                rec[k] = np.array([r[k] for r in runs], np.float64)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        out[name] = {k: [float(x) for x in rec[k]] for k in ("n_clusters", "n_big", "ari", "purity_big", "genomes_recovered")}
        out[name]["loss_last"] = [float(x) for x in rec["losses"][:, -1, 0]]
        print("e2e", name, out[name])
    return out


def main():
    which = sys.argv[1:] or ["cluster", "prep", "vae"]
    import torch

    torch.set_num_threads(1)  # deterministic MKL reductions for the stored vectors
    vt, cl, en = ref_harness.load_reference()
    manifest_path = os.path.join(HERE, "golden_manifest.json")
    manifest = json.load(open(manifest_path)) if os.path.exists(manifest_path) else {}
    manifest["environment"] = dict(torch=torch.__version__, numpy=np.__version__, threads=1,
                                   reference="RasmussenLab/vamb @ /root/reference (v5.0.x)")
    if "cluster" in which:
        manifest["cluster"] = gen_cluster(cl)
    if "cluster_large" in which:   # 100 k-point streams (minutes of reference CPU time each)
        manifest["cluster_large"] = gen_cluster(cl, fd.CLUSTER_CASES_LARGE)
    if "cluster_large_defined_order" in which:
        # The same inputs through the defined-order restatement (oracle/cluster_oracle.py + cluster_scan.c), stored
        # because at this size the REAL reference is no longer reproducible bit for bit by anything but itself: see
        # oracle/analyze_near_tie.py / profiles/r02_near_tie_100k_s050.txt.  The GPU must equal THIS stream exactly and
        # the reference stream up to the documented divergence.
        import cluster_oracle as co

        co.set_order(0)   # the ascending fmaf chain (libvambhip: scan.reference_order = 0); the oracle's default is the reference's order
        out = {}
        for name in fd.CLUSTER_CASES_LARGE:
            mat, lens, kw = fd.cluster_inputs(name)
            packed = fd.pack_stream(list(co.OracleClusterGenerator(mat.copy(), lens, **kw)))
            np.savez_compressed(os.path.join(HERE, f"cluster_{name}.defined_order.npz"), **packed)
            ref = fd.load("cluster_" + name)
            n = min(len(packed["medoid"]), len(ref["medoid"]))
            diff = np.flatnonzero(packed["medoid"][:n] != ref["medoid"][:n])
            out[name] = dict(n_clusters=len(packed["medoid"]),
                             identical_prefix_with_reference=int(diff[0]) if len(diff) else n)
            print("defined-order", name, out[name])
        co.set_order(co.DEFAULT_ORDER)
        manifest["cluster_large_defined_order"] = out
    if "tnf" in which:
        manifest["tnf"] = gen_tnf()
    if "prep" in which:
        manifest["prep"] = gen_prep(en)
    if "vae" in which:
        manifest["vae"] = gen_vae(en)
    for w in which:   # one case only: `make_golden.py vae:vae_c2_shape` (the other fixtures of the family stay as they are)
        if w.startswith("vae:"):
            manifest.setdefault("vae", {}).update(gen_vae(en, only=w[4:].split(",")))
    if "semisup" in which:
        manifest["semisup"] = gen_semisup()
    if "vaevae" in which:
        manifest["vaevae"] = gen_vaevae(en)
    if "e2e" in which:   # minutes of reference CPU time; 8 threads, free-running RNG
        manifest["e2e"] = gen_e2e()
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
