"""The numpy restatement of VAEConcat / VAELabels (oracle/semisup_oracle.py: explicit backward, restated D-Adapt-Adam,
torch Adam) against golden vectors recorded from the REAL reference classes
(/root/reference/vamb/semisupervised_encode.py:189,438 under torch autograd; tests/golden/make_golden.py semisup)."""
import numpy as np
import pytest

import fixture_defs as fd
import semisup_oracle as so
import vae_oracle as vo


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rows_of(name, g, lo, hi):
    """Input rows [lo, hi) in the model's column order, from the tensors the reference's own loader produced."""
    c = fd.SEMISUP_CASES[name]
    onehot = np.eye(fd.semisup_width(name), dtype=np.float32)[g["labels"][lo:hi]]
    if c["kind"] == "labels":
        return onehot
    return np.concatenate([g["depths"][lo:hi], g["tnf"][lo:hi], g["total_abundance"][lo:hi], onehot], axis=1)


def make_oracle(name, g, dtype=np.float64):
    c = fd.SEMISUP_CASES[name]
    NL = fd.semisup_width(name)
    width = NL if c["kind"] == "labels" else c["nsamples"] + 104 + NL
    st0 = vo.init_state(0, c["nhiddens"], c["nlatent"], c["seed"], width=width)
    return so.OracleLabelVAE(c["kind"], c["nsamples"], NL, c["nhiddens"], c["nlatent"], float(g["alpha"]), c["beta"],
                             c["dropout"], state=st0, dtype=dtype)


@pytest.mark.parametrize("name", list(fd.SEMISUP_CASES))
def test_oracle_matches_reference(name):
    c = fd.SEMISUP_CASES[name]
    g = fd.load(name)
    masks, eps = fd.semisup_randomness(name)
    B = c["batch"]
    m = make_oracle(name, g)
    x = rows_of(name, g, 0, B)
    w = g["weights"][:B] if c["kind"] == "concat" else None
    S = c["nsamples"]
    for step in range(c["steps"]):
        out, mu = m.forward_rows(x, eps=eps[step], masks=masks[step], train=True)
        res = m.loss_and_backward(x, w)
        if step == 0:
            assert rel(mu, g["step0_mu"]) < 5e-6
            if c["kind"] == "concat":
                assert rel(out[:, :S], g["step0_depths_out"]) < 5e-6
                assert rel(out[:, S:S + 103], g["step0_tnf_out"]) < 5e-6
                assert rel(out[:, S + 103:S + 104], g["step0_ab_out"]) < 5e-6
                assert rel(out[:, S + 104:], g["step0_labels_out"]) < 5e-6
            else:
                assert rel(out, g["step0_labels_out"]) < 5e-6
            for n in m.names:
                assert rel(m.grads[n], g["grad0/" + n]) < 3e-5, n
        if c["kind"] == "labels":
            m.adam_step(c["lrate"])
        else:
            m.dadapt_step()
        loss, ce_raw, sse_raw, cel, kld_raw, correct = g["losses"][step]
        assert abs(res["loss"] - loss) < 3e-6 * abs(loss)
        assert abs(res["ce_labels"] - cel) < 3e-6 * abs(cel)
        assert res["correct"] == int(correct)
        kld_w = 1 / (c["nlatent"] * c["beta"])
        assert abs(res["kld"] / kld_w - kld_raw) < 1e-5 * abs(kld_raw)
    for k, v in m.state.items():
        if v.dtype.kind != "f":
            assert int(v) == int(g["final/" + k])
        else:
            # Adam divides every element by its own running magnitude: elements whose gradient is small against its fp32
            # rounding error move by a sizeable fraction of lr in either direction (2e-5 at lr 1e-3, 1e-4 at lr 1e-2 here)
            assert rel(v, g["final/" + k]) < (3e-2 * c["lrate"] if c["kind"] == "labels" else 5e-6), k
    lat = m.encode_rows(rows_of(name, g, 0, c["n"]))
    assert (lat.view(np.uint32) & 0xFFF == 0).all()
    assert np.abs(lat - g["latent"]).max() <= np.abs(g["latent"]).max() * 2.0 ** -10


def test_label_cross_entropy_is_a_batch_mean_scaled_by_the_mean_weight():
    """semisupervised_encode.py:553-562: `ce_labels` is a scalar added to every row before the [B] x [B,1] broadcast with the
    weights, so loss.mean() = (mean of the row terms + ce_labels) * mean(weights)."""
    name = "semisup_concat_wide"
    c = fd.SEMISUP_CASES[name]
    g = fd.load(name)
    masks, eps = fd.semisup_randomness(name)
    m = make_oracle(name, g)
    x = rows_of(name, g, 0, c["batch"])
    w = g["weights"][:c["batch"]]
    m.forward_rows(x, eps=eps[0], masks=masks[0])
    r = m.loss_and_backward(x, w)
    assert abs(r["loss"] - (r["ab"] + r["ce"] + r["sse"] + r["ce_labels"] + r["kld"]) * w.astype(np.float64).mean()) < 1e-12
