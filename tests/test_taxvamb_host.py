"""Host side of the TaxVamb training mirror (vamb_amd/taxvamb_encode.py; no GPU): the taxonomy graph, the leaf masks and the
flat-softmax loss against the REAL reference functions (/root/reference/vamb/taxvamb_encode.py:29-61, vamb/hloss_misc.py) where
the reference tree is present, against the oracle otherwise; the loaders against the reference's loaders."""
import numpy as np
import pytest
import torch

import fixture_defs as fd
import ref_harness
import vaevae_oracle as vv
from vamb_amd import taxvamb_encode as vt

needs_reference = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


class Tax:
    def __init__(self, ranks):
        self.ranks = ranks


def random_taxonomies(seed, n=300):
    rng = np.random.RandomState(seed)
    phyla = [f"p{i}" for i in range(4)]
    out = []
    for _ in range(n):
        u = rng.random_sample()
        if u < 0.05:
            out.append(None)
            continue
        if u < 0.1:
            out.append(Tax([]))
            continue
        p = phyla[rng.randint(4)]
        c = f"{p}_c{rng.randint(3)}"
        o = f"{c}_o{rng.randint(2)}"
        depth = rng.randint(1, 5)
        out.append(Tax((["d_Bacteria", p, c, o])[:depth]))
    return out


def test_make_graph_orders_nodes_breadth_first():
    taxes = [Tax(["d", "p1", "c1"]), None, Tax([]), Tax(["d", "p2"]), Tax(["d", "p1", "c2"]), Tax(["d", "p2", "c3"])]
    nodes, ind, parents = vt.make_graph(taxes)
    assert nodes == ["root", "d", "p1", "p2", "c1", "c2", "c3"]
    assert parents == [-1, 0, 1, 1, 2, 2, 3]
    assert ind["c3"] == 6
    with pytest.raises(ValueError):    # a taxon under two parents (the reference's `only`, taxvamb_encode.py:64-71)
        vt.make_graph([Tax(["d", "p1", "x"]), Tax(["d", "p2", "x"])])


@needs_reference
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_make_graph_equals_the_reference(seed):
    tx = ref_harness.load_reference_module("taxvamb_encode")
    taxes = random_taxonomies(seed)
    assert vt.make_graph(taxes) == tx.make_graph(taxes)


def test_leaf_masks_equal_the_oracle():
    for name in fd.VAEVAE_CASES:
        parents = fd.vaevae_tree(name)
        assert np.array_equal(vt.leaf_masks(parents), vv.leaf_masks_of_nodes(parents))
    assert vt.leaf_masks([-1]).tolist() == [[True]]
    for bad in ([], [0], [-1, 1], [-1, 0, 5], [-1, -1]):
        with pytest.raises(ValueError):
            vt.leaf_masks(bad)


@needs_reference
def test_flat_softmax_equals_the_reference():
    hl = ref_harness.load_reference_module("hloss_misc")
    for name in fd.VAEVAE_CASES:
        parents = fd.vaevae_tree(name)
        tree = hl.Hierarchy(parents)
        ref = hl.FlatSoftmaxNLL(tree)
        mine = vt.FlatSoftmaxNLL(parents)
        assert torch.equal(mine.leaf_masks, ref.leaf_masks)
        rng = np.random.RandomState(3)
        n_leaves = mine.leaf_masks.shape[1]
        scores = torch.from_numpy((3 * rng.standard_normal((50, n_leaves))).astype(np.float32))
        onehot = torch.nn.functional.one_hot(torch.from_numpy(rng.randint(0, len(parents), size=50)), max(len(parents), 105)).float()
        assert torch.equal(mine(scores, onehot), ref(scores, onehot))
        assert vt.init_hier_loss("flat_softmax", parents).n_labels == n_leaves
    with pytest.raises(NotImplementedError):
        vt.init_hier_loss("cond_softmax", parents)
    with pytest.raises(AttributeError):
        vt.init_hier_loss("nope", parents)


@needs_reference
def test_loaders_yield_the_reference_tensors_and_batches():
    """make_dataloader_{labels,concat,semisupervised}_hloss: same dataset tensors (the seeded permutations included), same batch
    size / drop_last / sampler kind, same collated batches as the reference's loaders on the same inputs."""
    _, _, en = ref_harness.load_reference()
    tx = ref_harness.load_reference_module("taxvamb_encode")
    from vamb_amd import encode as ve

    name = "vaevae_tree_wide"
    c = fd.VAEVAE_CASES[name]
    ab, tnf, lens, nodes, parents = fd.vaevae_inputs(name)
    N, B, S = len(parents), c["batch"], c["nsamples"]
    ve.set_prep_mode("host")
    try:
        mine = (vt.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B),
                ve.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B),
                vt.make_dataloader_labels_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B))
    finally:
        ve.set_prep_mode("auto")
    ref = (tx.make_dataloader_concat_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B),
           en.make_dataloader(ab.copy(), tnf.copy(), lens, batchsize=B),
           tx.make_dataloader_labels_hloss(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=B))
    for a, b in zip(mine, ref):
        assert a.batch_size == b.batch_size and a.drop_last == b.drop_last and type(a.sampler) is type(b.sampler)
        assert len(a.dataset.tensors) == len(b.dataset.tensors)
        for x, y in zip(a.dataset.tensors, b.dataset.tensors):
            assert x.dtype == y.dtype and torch.equal(x, y)
    m_all = vt.make_dataloader_semisupervised_hloss(*mine, N, parents, (S, 103, 1, N), c["perm_seed"], batchsize=B)
    r_all = tx.make_dataloader_semisupervised_hloss(*ref, N, parents, (S, 103, 1, N), c["perm_seed"], batchsize=B)
    assert m_all.batch_size == r_all.batch_size and m_all.drop_last == r_all.drop_last
    assert type(m_all.sampler) is type(r_all.sampler) is torch.utils.data.SequentialSampler
    for x, y in zip(m_all.dataset.tensors, r_all.dataset.tensors):
        assert torch.equal(x, y)
    g = fd.load(name)   # ... which are the tensors the golden run trained on
    assert torch.equal(m_all.dataset.tensors[0], torch.from_numpy(g["unsup_depths"]))
    assert torch.equal(m_all.dataset.tensors[9], torch.from_numpy(g["sup_nodes"]))
    for mb, rb in zip(m_all, r_all):
        assert len(mb) == len(rb) == 10
        for x, y in zip(mb, rb):
            assert torch.equal(x, y)
    assert np.array_equal(vt.permute_indices(10, 25, 1), tx.permute_indices(10, 25, 1))
    with pytest.raises(ValueError):
        vt.make_dataloader_labels_hloss(ab, tnf[:-1], lens, nodes, N, parents)


def test_labels_loader_validates_and_destroys_like_the_reference():
    """taxvamb_encode.make_dataloader_labels_hloss builds the FEATURE dataset first (`_make_dataset`) and throws it
    away, so the labels-only loader inherits make_dataloader's validation (dtypes, shapes, zero rows, batch size against the
    dataset) and, with destroy=True, normalises the caller's arrays in place.  Same inputs through both."""
    tx = ref_harness.load_reference_module("taxvamb_encode")
    name = "vaevae_tree_drop"
    c = fd.VAEVAE_CASES[name]
    ab, tnf, lens, nodes, parents = fd.vaevae_inputs(name)
    N, B = len(parents), c["batch"]
    bad_inputs = [
        (ab.astype(np.float64), tnf, lens),                       # abundance not float32
        (ab, tnf.astype(np.float64), lens),                       # TNF not float32
        (ab, tnf[:, :50], lens),                                  # TNF of the wrong width
        (ab[:, :0], tnf, lens),                                   # no samples
    ]
    zero = ab.copy()
    zero[:, 1] = 0.0                                              # a sample nobody has any depth in
    bad_inputs.append((zero, tnf, lens))
    verdicts = []
    for a, t, ln in bad_inputs:
        outcome = []
        for fn in (tx.make_dataloader_labels_hloss, vt.make_dataloader_labels_hloss):
            try:
                fn(a.copy(), t.copy(), ln, nodes, N, parents, batchsize=B)
                outcome.append(None)
            except Exception as e:   # noqa: BLE001 -- the point is that both sides fail (or not) with the same exception type
                outcome.append(type(e))
        assert outcome[0] is outcome[1], outcome
        verdicts.append(outcome[0])
    assert verdicts[0] is ValueError and verdicts[1] is ValueError and verdicts[-1] is ValueError   # dtypes, the empty sample
    for fn in (tx.make_dataloader_labels_hloss, vt.make_dataloader_labels_hloss):
        with pytest.raises(ValueError):
            fn(ab.copy(), tnf.copy(), lens, nodes, N, parents, batchsize=0)
    # destroy=True: both normalise the caller's arrays in place, to the same values
    a_ref, t_ref = ab.copy(), tnf.copy()
    a_mine, t_mine = ab.copy(), tnf.copy()
    r = tx.make_dataloader_labels_hloss(a_ref, t_ref, lens, nodes, N, parents, batchsize=B, destroy=True)
    m = vt.make_dataloader_labels_hloss(a_mine, t_mine, lens, nodes, N, parents, batchsize=B, destroy=True)
    assert not np.array_equal(a_ref, ab) and not np.array_equal(t_ref, tnf)
    assert np.array_equal(a_ref, a_mine) and np.array_equal(t_ref, t_mine)
    assert torch.equal(r.dataset.tensors[0], m.dataset.tensors[0]) and r.batch_size == m.batch_size and r.drop_last == m.drop_last
    # destroy=False leaves them alone
    a2, t2 = ab.copy(), tnf.copy()
    vt.make_dataloader_labels_hloss(a2, t2, lens, nodes, N, parents, batchsize=B)
    assert np.array_equal(a2, ab) and np.array_equal(t2, tnf)
