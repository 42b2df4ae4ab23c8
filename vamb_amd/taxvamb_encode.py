"""TaxVamb's models on MI355X -- drop-in for the training side of ``/root/reference/vamb/taxvamb_encode.py`` (SURVEY.md 8f
row N4): the loaders (:74-239), ``VAELabelsHLoss`` (:277-419), ``VAEConcatHLoss`` (:422-538) and the joint trainer
``VAEVAEHLoss`` (:551-743), the model ``vamb bin taxvamb`` trains (``__main__.py:1988-2047``).

The labels are NODES of a taxonomy given as a parent table in BFS order (``make_graph``, :29-61).  The hierarchical loss is the
reference's default ``flat_softmax`` (``hloss_misc.FlatSoftmaxNLL``, hloss_misc.py:1102-1133): -log of the softmax mass -- over
the first ``n_leaves`` label logits -- on the leaves at or below the contig's node.  It runs inside the fused loss kernel of
libvambhip (``vh_vae_set_hierarchy``); ``cond_softmax`` / ``soft_margin`` (selectable in the reference's constructor, never by
its command line) are not implemented and raise.  Out of scope here: ``VAMB2Label`` (Taxometer's classifier, :746-1106) and the
prediction helpers the HLoss classes carry for it (``pred_helper``, ``find_lca``, ``eval_label_map``).
"""
from __future__ import annotations

import ctypes
from collections import namedtuple
from functools import partial
from math import log as _log
from typing import Optional, Sequence

import numpy as _np
import torch as _torch
from torch.utils.data import DataLoader as _DataLoader
from torch.utils.data.dataset import TensorDataset as _TensorDataset

from . import _lib
from . import encode as _encode
from . import semisupervised_encode as _semisupervised_encode
from .encode import logger

DEFAULT_HIER_LOSS = "flat_softmax"
HierLoss = namedtuple("HierLoss", ["name", "loss_fn", "pred_helper", "pred_fn", "n_labels"])


# ---- taxonomy graph (taxvamb_encode.py:29-71) -------------------------------------------------------------------------
def only(itr):
    itr = iter(itr)
    y = next(itr)
    try:
        next(itr)
    except StopIteration:
        return y
    raise ValueError("More than one element in iterator")


def make_graph(taxes: Sequence) -> tuple[list[str], dict[str, int], list[int]]:
    """Nodes in breadth-first order from "root" (children in the order their edges were first seen, as networkx's
    ``bfs_edges`` walks its insertion-ordered adjacency), their indices, and the parent of every node (-1 for the root).
    ``taxes``: per contig an object with a ``ranks`` list (``vamb.taxonomy.ContigTaxonomy``) or None."""
    logger.info("Creating taxonomy graph from contig taxonomies")
    root = "root"
    children: dict[str, list[str]] = {root: []}
    parents: dict[str, list[str]] = {root: []}

    def add_edge(a, b):
        for n in (a, b):
            children.setdefault(n, [])
            parents.setdefault(n, [])
        if b not in children[a]:
            children[a].append(b)
            parents[b].append(a)

    for contig_taxonomy in taxes:
        if contig_taxonomy is None or len(contig_taxonomy.ranks) == 0:
            continue
        add_edge(root, contig_taxonomy.ranks[0])
        for parent, child in zip(contig_taxonomy.ranks, contig_taxonomy.ranks[1:]):
            add_edge(parent, child)
    nodes, seen, queue = [root], {root}, [root]
    while queue:
        nxt = []
        for u in queue:
            for v in children[u]:
                if v not in seen:
                    seen.add(v)
                    nodes.append(v)
                    nxt.append(v)
        queue = nxt
    ind_nodes = {v: i for i, v in enumerate(nodes)}
    assert len(ind_nodes) == len(nodes)
    table_parent: list[int] = []
    for n in nodes:
        if n == root:
            table_parent.append(-1)
        else:
            parent_index = ind_nodes[only(parents[n])]
            assert parent_index < ind_nodes[n]
            table_parent.append(parent_index)
    return nodes, ind_nodes, table_parent


def leaf_masks(table_parent) -> _np.ndarray:
    """[n_nodes][n_leaves] bool: leaf j (nodes nobody names as parent, in node order) is node i or below it
    (hloss_misc.py:51-58, 98-115, 1110-1113)."""
    p = [int(x) for x in table_parent]
    n = len(p)
    if n < 1 or p[0] != -1 or any(not (0 <= p[i] < i) for i in range(1, n)):
        raise ValueError("table_parent must start with -1 (the root) and name every other node's parent before the node")
    is_leaf = _np.ones(n, bool)
    for i in range(1, n):
        is_leaf[p[i]] = False
    leaf_of = _np.cumsum(is_leaf) - 1
    m = _np.zeros((n, int(is_leaf.sum())), bool)
    for j in _np.flatnonzero(is_leaf):
        i = int(j)
        while i >= 0:
            m[i, leaf_of[j]] = True
            i = p[i]
    return m


class FlatSoftmaxNLL:
    """``hloss_misc.FlatSoftmaxNLL(tree)`` on host tensors (hloss_misc.py:1102-1133): cross-entropy that supports internal
    labels.  ``labels`` are one-hot rows over the nodes (any width >= n_nodes), ``scores`` the n_leaves leaf logits."""

    def __init__(self, table_parent, reduction: str = "mean"):
        assert reduction in ("mean", "none", None)
        self.leaf_masks = _torch.from_numpy(leaf_masks(table_parent))
        self.reduction = reduction

    def __call__(self, scores, labels):
        labels = _torch.argmax(labels, dim=1)
        logp_leaf = _torch.nn.functional.log_softmax(scores, dim=-1)
        label_leaf_mask = self.leaf_masks[labels.long(), :]
        logp_ancestors = _torch.where(label_leaf_mask, logp_leaf, _torch.tensor(-_torch.inf))
        loss = -_torch.logsumexp(logp_ancestors, dim=-1)
        return _torch.mean(loss) if self.reduction == "mean" else loss

    forward = __call__


def init_hier_loss(name, table_parent):
    """taxvamb_encode.py:242-274, restricted to the loss the product implements."""
    if name == "flat_softmax":
        m = leaf_masks(table_parent)
        return HierLoss(name="flat_softmax", loss_fn=FlatSoftmaxNLL(table_parent), pred_helper=None, pred_fn=None,
                        n_labels=m.shape[1])
    if name in ("cond_softmax", "soft_margin"):
        raise NotImplementedError(f"hierarchical loss {name!r} is not implemented on the GPU path (only the default "
                                  f"{DEFAULT_HIER_LOSS!r})")
    raise AttributeError(f"Hierarchical loss {name} not found")


# ---- loaders (taxvamb_encode.py:74-239) -------------------------------------------------------------------------------
def collate_fn_labels_hloss(num_categories: int, table_parent, labels):
    return _semisupervised_encode.collate_fn_labels(num_categories, labels)


def collate_fn_concat_hloss(num_categories: int, table_parent, batch):
    a = _torch.stack([i[0] for i in batch])
    b = _torch.stack([i[1] for i in batch])
    c = _torch.stack([i[2] for i in batch])
    d = _torch.stack([i[3] for i in batch])
    e = [i[4] for i in batch]
    return a, b, c, d, collate_fn_labels_hloss(num_categories, table_parent, e)[0]


def collate_fn_semisupervised_hloss(num_categories: int, table_parent, batch):
    return _semisupervised_encode.collate_fn_semisupervised(num_categories, batch)


def make_dataloader_labels_hloss(rpkm, tnf, lengths, labels, N, table_parent, batchsize=256, destroy=False, cuda=False):
    """taxvamb_encode.py:114-139: one node index per contig.  The reference builds the feature dataset first (``_make_dataset``)
    and throws it away: the feature arrays contribute their validation (dtypes, shapes, batch size against the dataset,
    zero-depth / zero-TNF rows) and, with ``destroy=True``, are normalised in place -- both kept by routing them through
    ``make_dataloader`` (host path: nothing of it is used afterwards)."""
    _encode.make_dataloader(rpkm, tnf, lengths, batchsize, destroy, cuda, _prep="host")
    dataset = _TensorDataset(_torch.Tensor(labels).long())
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=dataset.tensors[0].shape[0] > batchsize, shuffle=True,
                       num_workers=0, pin_memory=False, collate_fn=partial(collate_fn_labels_hloss, N, table_parent))


def make_dataloader_concat_hloss(rpkm, tnf, lengths, labels, N: int, table_parent: list[int], no_filter: bool = True,
                                 batchsize: int = 256, destroy: bool = False, cuda: bool = False):
    """taxvamb_encode.py:142-178: the four tensors of ``make_dataloader`` + the node index of every contig."""
    base = _encode.make_dataloader(rpkm, tnf, lengths, batchsize, destroy, cuda)
    tensors = base.dataset.tensors
    dataset = _TensorDataset(*tensors, _torch.Tensor(labels).long())
    prepared = getattr(base.dataset, "_vambhip_prepared", None)
    if prepared is not None:          # features normalised on the device: the resident matrix is shared, not copied
        dataset._vambhip_prepared = prepared
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=len(tensors[0]) > batchsize, shuffle=True, num_workers=0,
                       pin_memory=False, collate_fn=partial(collate_fn_concat_hloss, N, table_parent))


def permute_indices(n_current: int, n_total: int, seed: int):
    """taxvamb_encode.py:181-189."""
    rng = _np.random.default_rng(seed)
    x = _np.arange(n_current)
    to_add = n_total // n_current
    to_concatenate = [rng.permutation(x)]
    for _ in range(to_add):
        to_concatenate.append(rng.permutation(x))
    return _np.concatenate(to_concatenate)[:n_total]


def make_dataloader_semisupervised_hloss(dataloader_joint, dataloader_vamb, dataloader_labels, N, table_parent, shapes, seed: int,
                                         batchsize=256, cuda=False):
    """taxvamb_encode.py:192-239: ten row-aligned tensors -- a seeded permutation of the unsupervised features, of the
    unsupervised labels and of the supervised rows -- served in order (shuffle=False)."""
    n_total = len(dataloader_vamb.dataset)
    indices_unsup_vamb = permute_indices(len(dataloader_vamb.dataset), n_total, seed)
    indices_unsup_labels = permute_indices(len(dataloader_labels.dataset), n_total, seed)
    indices_sup = permute_indices(len(dataloader_joint.dataset), n_total, seed)
    tv, tl, tj = dataloader_vamb.dataset.tensors, dataloader_labels.dataset.tensors, dataloader_joint.dataset.tensors
    dataset_all = _TensorDataset(tv[0][indices_unsup_vamb], tv[1][indices_unsup_vamb], tv[2][indices_unsup_vamb],
                                 tv[3][indices_unsup_vamb], tl[0][indices_unsup_labels], tj[0][indices_sup], tj[1][indices_sup],
                                 tj[2][indices_sup], tj[3][indices_sup], tj[4][indices_sup])
    return _DataLoader(dataset=dataset_all, batch_size=batchsize, drop_last=len(indices_unsup_vamb) > batchsize, shuffle=False,
                       num_workers=0, pin_memory=False, collate_fn=partial(collate_fn_semisupervised_hloss, N, table_parent))


# ---- models ---------------------------------------------------------------------------------------------------------------
class _HLossMixin:
    """What the two single-network HLoss classes add to their one-hot parents: the taxonomy, resident on the device."""

    def _init_hierarchy(self, nodes, table_parent, hier_loss):
        self.nodes = nodes
        self.table_parent = table_parent
        self.hierloss = init_hier_loss(hier_loss, table_parent)
        self.nlabels = self.hierloss.n_labels          # taxvamb_encode.py:329 / 477: from here on the number of LEAVES
        self.loss_fn = self.hierloss.loss_fn
        # the hierarchical loss is an fp32-step kernel
        if self.compute_dtype != "fp32":
            _lib.check(self._lib.vh_vae_set_precision(self._h, 0))
            self.compute_dtype = "fp32"
        tp = _np.ascontiguousarray(table_parent, dtype=_np.int32)
        _lib.check(self._lib.vh_vae_set_hierarchy(self._h, _lib.ptr(tp), len(tp)))


class VAELabelsHLoss(_HLossMixin, _semisupervised_encode.VAELabels):
    """Variational autoencoder that encodes only the labels; the labels are nodes of a taxonomy and the reconstruction loss
    is hierarchical (taxvamb_encode.py:277-419).
        nlabels: width of the label block;  nodes, table_parent: the taxonomy in BFS order;  the rest as VAELabels."""

    _OPTIMIZER = _semisupervised_encode.VH_OPT_DADAPT_ADAM   # `dadaptation.DAdaptAdam(lr=1, decouple=True)` (:386)

    def __init__(self, nlabels: int, nodes, table_parent, nhiddens=None, nlatent: int = 32, alpha: Optional[float] = None,
                 beta: float = 200.0, dropout: Optional[float] = 0.2, hier_loss=DEFAULT_HIER_LOSS, cuda: bool = False, _seed: int = 0):
        super().__init__(nlabels, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout, cuda=cuda, _seed=_seed)
        self._init_hierarchy(nodes, table_parent, hier_loss)

    def calc_loss(self, labels_in, labels_out, mu, logsigma):
        """taxvamb_encode.py:348-355 on host tensors."""
        t = lambda x: x if isinstance(x, _torch.Tensor) else _torch.as_tensor(x)  # noqa: E731
        labels_in, labels_out, mu, logsigma = t(labels_in), t(labels_out), t(mu), t(logsigma)
        ce_labels = self.loss_fn(labels_out, labels_in)
        kld = -0.5 * (1 + logsigma - mu.pow(2) - logsigma.exp()).sum(dim=1).mean()
        loss = ce_labels * 1.0 + kld * (1 / (self.nlatent * self.beta))
        return loss, ce_labels, kld, _torch.tensor(0)

    def trainmodel(self, dataloader, nepochs: int = 500, lrate: float = 1e-3, batchsteps: Optional[list[int]] = [25, 75, 150, 300],
                   modelfile=None):
        """taxvamb_encode.py:357-419: VAELabels.trainmodel with D-Adapt-Adam instead of Adam (``lrate`` is only logged)."""
        return self._trainmodel(dataloader, nepochs, lrate, batchsteps, modelfile, _semisupervised_encode.VH_OPT_DADAPT_ADAM)


class VAEConcatHLoss(_HLossMixin, _semisupervised_encode.VAEConcat):
    """Variational autoencoder on the concatenated input of VAMB and labels, hierarchical label loss
    (taxvamb_encode.py:422-538).  Trains with the inherited ``VAE.trainmodel`` (D-Adapt-Adam) like its parent."""

    def __init__(self, nsamples: int, nlabels: int, nodes, table_parent, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha=None, beta: float = 200.0, dropout: Optional[float] = 0.2, hier_loss=DEFAULT_HIER_LOSS, cuda: bool = False,
                 _seed: int = 0):
        super().__init__(nsamples, nlabels, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout, cuda=cuda,
                         _seed=_seed)
        self._init_hierarchy(nodes, table_parent, hier_loss)

    def calc_loss(self, depths_in, depths_out, tnf_in, tnf_out, abundance_in, abundance_out, labels_in, labels_out, mu, logsigma,
                  weights):
        """taxvamb_encode.py:496-538 on host tensors."""
        t = lambda x: x if isinstance(x, _torch.Tensor) else _torch.as_tensor(x)  # noqa: E731
        depths_in, depths_out, tnf_in, tnf_out = t(depths_in), t(depths_out), t(tnf_in), t(tnf_out)
        abundance_in, abundance_out, labels_in, labels_out = t(abundance_in), t(abundance_out), t(labels_in), t(labels_out)
        mu, weights = t(mu), t(weights)
        ab_sse = (abundance_out - abundance_in).pow(2).sum(dim=1)
        ce = -((depths_out + 1e-9).log() * depths_in).sum(dim=1)
        sse = (tnf_out - tnf_in).pow(2).sum(dim=1)
        kld = 0.5 * (mu.pow(2)).sum(dim=1)
        ce_weight = 0.0 if self.nsamples == 1 else ((1 - self.alpha) * (self.nsamples - 1)) / (self.nsamples * _log(self.nsamples))
        ab_sse_weight = (1 - self.alpha) * (1 / self.nsamples)
        sse_weight = self.alpha / self.ntnf
        kld_weight = 1 / (self.nlatent * self.beta)
        ce_labels = self.loss_fn(labels_out, labels_in)
        loss = (ab_sse * ab_sse_weight + ce * ce_weight + sse * sse_weight + ce_labels * 1.0 + kld * kld_weight) * weights
        return loss, ce, sse, ce_labels, kld, _torch.tensor(0)


def kld_gauss(p_mu, p_logstd, q_mu, q_logstd):
    """taxvamb_encode.py:541-548."""
    return _semisupervised_encode.kld_gauss(p_mu, p_logstd, q_mu, q_logstd)


class VAEVAEHLoss(_semisupervised_encode.VAEVAE):
    """Bi-modal variational autoencoder that uses TNFs, abundances and labels: three encoders (VAMB, labels, concatenated
    VAMB + labels), two decoders (VAMB and labels), hierarchical loss over the taxonomy (taxvamb_encode.py:551-743).
        nsamples, nlabels (= number of nodes), nodes, table_parent; hyperparameters as VAE.
    vae.trainmodel(dataloader, nepochs, lrate, batchsteps, modelfile) trains the three networks jointly
    (``make_dataloader_semisupervised_hloss``); ``vae.VAEJoint.encode(dataloader_joint)`` gives the latent TaxVamb clusters."""

    def __init__(self, nsamples: int, nlabels: int, nodes, table_parent, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha: Optional[float] = None, beta: float = 200.0, dropout: Optional[float] = 0.2, hier_loss=DEFAULT_HIER_LOSS,
                 cuda: bool = False):
        self.usecuda = cuda
        N_l = max(nlabels, 105)
        kw = dict(nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout, cuda=cuda)
        # (distinct seeds: the three networks draw independent dropout / noise / initialisation streams)
        self.VAEVamb = _encode.VAE(nsamples, seed=0, **kw)
        self.VAELabels = VAELabelsHLoss(N_l, nodes, table_parent, hier_loss=hier_loss, _seed=1, **kw)
        self.VAEJoint = VAEConcatHLoss(nsamples, N_l, nodes, table_parent, hier_loss=hier_loss, _seed=2, **kw)

    def _label_loss(self, labels_out, labels_in):
        return self.VAEJoint.loss_fn(labels_out, labels_in), _torch.tensor(0)

    @classmethod
    def load(cls, path, nodes, table_parent, cuda=False, evaluate=True):
        """taxvamb_encode.py:630-680.  ``save`` stores ``VAELabels.nlabels`` under "nlabels" as the reference does -- and that is
        the number of LEAVES (the HLoss classes overwrite it after the layers were built max(n_nodes, 105) wide), so the
        reference's own round trip breaks for a taxonomy of more than 105 nodes (load_state_dict size mismatch).  The file format
        stays the reference's; the label block's width is taken from the taxonomy passed in (``table_parent``: one entry per
        node) and checked against the stored weights."""
        d = _torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
        n_nodes = len(table_parent)
        width = max(n_nodes, 105)
        stored = d["state_VAELabels"]["encoderlayers.0.weight"].shape[1]
        if stored != width:
            raise ValueError(f"the model was trained with a label block of {stored} columns, the taxonomy passed to load has "
                             f"{n_nodes} nodes (a block of {width})")
        vae = cls(d["nsamples"], n_nodes, nodes, table_parent, d["nhiddens"], d["nlatent"], d["alpha"], d["beta"],
                  d["dropout"], cuda=cuda)
        vae.VAEVamb.load_state_dict(d["state_VAEVamb"])
        vae.VAELabels.load_state_dict(d["state_VAELabels"])
        vae.VAEJoint.load_state_dict(d["state_VAEJoint"])
        if evaluate:
            for net in vae._networks():
                net.eval()
        return vae
