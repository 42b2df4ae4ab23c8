"""The numpy restatement of the joint TaxVamb trainer (oracle/vaevae_oracle.py: hierarchy masks, flat-softmax NLL, the seven
passes of VAEVAE.trainepoch with explicit backward, torch Adam) against golden vectors recorded by running the REAL
``VAEVAEHLoss.trainmodel`` (/root/reference/vamb/taxvamb_encode.py:551-743, semisupervised_encode.py:700-1084) under torch
autograd with injected dropout masks and noise (tests/golden/make_golden.py vaevae)."""
import numpy as np
import pytest

import fixture_defs as fd
import vae_oracle as vo
import vaevae_oracle as vv


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def init_states(name):
    c = fd.VAEVAE_CASES[name]
    S = c["nsamples"]
    NL = max(len(fd.vaevae_tree(name)), 105)
    widths = dict(VAEVamb=None, VAELabels=NL, VAEJoint=S + 104 + NL)
    return {k: vo.init_state(S if k != "VAELabels" else 0, c["nhiddens"], c["nlatent"], c["seed"] + i, width=w)
            for i, (k, w) in enumerate(widths.items())}


def make_oracle(name, g, dtype=np.float64):
    c = fd.VAEVAE_CASES[name]
    return vv.OracleVAEVAE(c["nsamples"], [int(p) for p in g["parents"]], c["nhiddens"], c["nlatent"], float(g["alpha"]), c["beta"],
                           c["dropout"], init_states(name), dtype=dtype)


def batch_of(g, which, lo, hi):
    return (dict(depths=g[which + "_depths"][lo:hi], tnf=g[which + "_tnf"][lo:hi], abundance=g[which + "_abundance"][lo:hi],
                 weights=g[which + "_weights"][lo:hi]), g[which + "_nodes"][lo:hi])


def step_batches(g, lo, hi):
    """(unsup, unsup_nodes, sup, sup_nodes) of rows [lo, hi) as VAEVAE.trainepoch binds them: BY POSITION
    (semisupervised_encode.py:864-875).  The first five tensors of the loader -- dataloader_vamb's features + dataloader_labels'
    labels, which make_dataloader_semisupervised_hloss (and therefore the fixtures' keys) call "unsup" -- are trainepoch's `*_sup`
    batch; the last five -- dataloader_joint, "sup" in the fixtures -- its `*_unsup` batch.  The fixture vaevae_tree_split, whose
    halves differ, is what tells this binding from the opposite one."""
    su, su_nodes = batch_of(g, "unsup", lo, hi)
    un, un_nodes = batch_of(g, "sup", lo, hi)
    return un, un_nodes, su, su_nodes


def test_leaf_masks_of_the_reference_test_taxonomy():
    """root -> domain -> 3 phyla -> 3 classes each (test/test_semisupervised_encode.py:22-30): 9 leaves; the root and the domain
    cover all of them, a phylum its three classes, a class itself."""
    parents = fd.vaevae_tree("vaevae_tree_drop")
    M = vv.leaf_masks_of_nodes(parents)
    assert M.shape == (14, 9)
    assert M[0].all() and M[1].all()
    for phylum in (2, 3, 4):
        assert M[phylum].sum() == 3
    for leaf in range(5, 14):
        assert M[leaf].sum() == 1 and M[leaf, leaf - 5]
    # loss of a root label is exactly zero, of a leaf label the plain cross-entropy
    rng = np.random.RandomState(0)
    sc = rng.standard_normal((4, 9))
    loss, d = vv.flat_softmax_nll(sc, np.array([0, 0, 1, 0]), M)
    assert abs(loss) < 1e-12 and np.abs(d).max() < 1e-12
    loss, _ = vv.flat_softmax_nll(sc, np.array([5, 6, 7, 13]), M)
    lsm = sc - np.log(np.exp(sc).sum(axis=1, keepdims=True))
    assert abs(loss + lsm[np.arange(4), [0, 1, 2, 8]].mean()) < 1e-12


@pytest.mark.parametrize("name", list(fd.VAEVAE_CASES))
def test_oracle_matches_reference(name):
    c = fd.VAEVAE_CASES[name]
    g = fd.load(name)
    rnd = fd.vaevae_randomness(name)
    B = c["batch"]
    m = make_oracle(name, g)
    assert m.n_leaves == int(np.sum(~np.isin(np.arange(len(g["parents"])), g["parents"])))
    gmax = {}   # largest gradient magnitude every element saw over the steps
    for step in range(c["steps"]):
        un, un_nodes, su, su_nodes = step_batches(g, step * B, (step + 1) * B)
        if step == 0:
            # gradients of step 0: run the step on a copy so that the real one below still starts from the initial weights
            probe = make_oracle(name, g)
            probe.train_step(un, un_nodes, su, su_nodes, rnd[0], lr=0.0)
            assert rel(probe.mu_sup, g["step0_mu_sup"]) < 5e-6
            for k, net in (("VAEVamb", probe.vamb), ("VAELabels", probe.labels), ("VAEJoint", probe.joint)):
                for n in net.names:
                    ref = g[f"grad0/{k}/{n}"]
                    # VAEJoint's decoder never receives a gradient (its outputs are discarded): exactly zero on both sides
                    if k == "VAEJoint" and (n.startswith("decoder") or n.startswith("outputlayer")):
                        assert not ref.any() and not net.grads[n].any(), n
                    else:
                        assert rel(net.grads[n], ref) < 5e-5, (k, n)
        got = m.train_step(un, un_nodes, su, su_nodes, rnd[step], lr=c["lrate"])
        ref = g["losses"][step]
        for i, key in enumerate(vv.METRICS):
            assert abs(got[i] - ref[i]) <= 5e-6 * abs(ref[i]) + 1e-9, (step, key, got[i], ref[i])
        for k, net in (("VAEVamb", m.vamb), ("VAELabels", m.labels), ("VAEJoint", m.joint)):
            for n in net.names:
                gmax[k, n] = np.maximum(gmax.get((k, n), 0.0), np.abs(net.grads[n]))
    for k, net in (("VAEVamb", m.vamb), ("VAELabels", m.labels), ("VAEJoint", m.joint)):
        for n, v in net.state.items():
            ref = g[f"final/{k}/{n}"]
            if v.dtype.kind != "f":
                assert int(v) == int(ref), (k, n)      # num_batches_tracked: 2 (encoder) / 3 (decoder) forwards per step
            elif (k, n) not in gmax:
                assert rel(v, ref) < 5e-6, (k, n)      # running statistics
            else:
                # Adam: elements whose gradient is small against its fp32 rounding error move by a sizeable fraction of lr;
                # elements whose gradient never left the neighbourhood of Adam's eps (1e-8: e.g. the weight of a one-hot
                # column whose only row has a vanishing dz) move by ~lr * g / (|g| + eps), i.e. by rounding noise
                solid = gmax[k, n] > 1e-6
                scale = np.abs(ref).max()
                assert not solid.any() or np.abs(v - ref)[solid].max() < (3e-2 * c["lrate"] + 5e-6) * scale, (k, n)
                assert np.abs(v - ref).max() <= c["steps"] * c["lrate"], (k, n)
    lat = m.encode_joint(g["joint_depths"], g["joint_tnf"], g["joint_abundance"], g["joint_nodes"])
    assert (lat.view(np.uint32) & 0xFFF == 0).all()
    assert np.abs(lat - g["latent_joint"]).max() <= np.abs(g["latent_joint"]).max() * 2.0 ** -9


def test_batches_tracked_counts_every_training_mode_forward():
    """Nobody switches the modules to eval inside VAEVAE.trainepoch: per step VAEVamb / VAELabels run their encoder twice and
    their decoder three times, VAEJoint both once (semisupervised_encode.py:899-928)."""
    name = "vaevae_tree_drop"
    g = fd.load(name)
    steps = fd.VAEVAE_CASES[name]["steps"]
    assert int(g["final/VAEVamb/encodernorms.0.num_batches_tracked"]) == 2 * steps
    assert int(g["final/VAEVamb/decodernorms.0.num_batches_tracked"]) == 3 * steps
    assert int(g["final/VAELabels/encodernorms.1.num_batches_tracked"]) == 2 * steps
    assert int(g["final/VAELabels/decodernorms.1.num_batches_tracked"]) == 3 * steps
    assert int(g["final/VAEJoint/encodernorms.0.num_batches_tracked"]) == steps
    assert int(g["final/VAEJoint/decodernorms.0.num_batches_tracked"]) == steps
