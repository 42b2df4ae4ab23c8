"""VAELabels / VAEConcat on MI355X -- drop-in for the two ``vamb.encode.VAE`` subclasses of
``/root/reference/vamb/semisupervised_encode.py`` (VAELabels :189-436, VAEConcat :438-698) and their loaders
(make_dataloader_labels :151-175, make_dataloader_concat :111-148).  SURVEY.md 8f row N4.

They are the same stack of layers as ``VAE`` on other input / reconstruction columns:

    VAEConcat   depths | TNF | total abundance | one-hot labels      loss = VAE loss + CrossEntropy(label logits)
    VAELabels   one-hot labels                                      loss = CrossEntropy + KLD / (nlatent * beta)

so they run on the kernels of ``vamb_amd.encode.VAE`` (``csrc/vae.hip``: the label block is one more segment of the fused
loss kernel; the one-hot columns are written by the batch gather from one int32 per row and never stored).  VAEConcat
inherits ``VAE.trainmodel`` (D-Adapt-Adam) exactly as in the reference; VAELabels trains with ``torch.optim.Adam(lr)``
semantics (its own trainmodel, :362-436), implemented in the same fused optimiser kernel.
"""
from __future__ import annotations

import ctypes
import weakref
from functools import partial
from math import log as _log
from typing import Optional

import numpy as _np
import torch as _torch
import torch.nn.functional as _F
from torch.utils.data import DataLoader as _DataLoader
from torch.utils.data.dataset import TensorDataset as _TensorDataset

from . import _lib
from . import encode as _encode
from .encode import logger

NTNF = _encode.NTNF
VH_VAE_CONCAT, VH_VAE_LABELS = 1, 2
VH_OPT_DADAPT_ADAM, VH_OPT_ADAM = 0, 1


class _LabelsConfig(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("nlabels", ctypes.c_int32), ("optimizer", ctypes.c_int32),
                ("lrate", ctypes.c_float)]


# ---- loaders (semisupervised_encode.py:25-175) ------------------------------------------------------------------------
def collate_fn_labels(num_categories: int, batch):
    return [_F.one_hot(_torch.as_tensor(batch), num_classes=max(num_categories, 105)).squeeze(1).float()]


def collate_fn_concat(num_categories: int, batch):
    a = _torch.stack([i[0] for i in batch])
    b = _torch.stack([i[1] for i in batch])
    c = _torch.stack([i[2] for i in batch])
    d = _torch.stack([i[3] for i in batch])
    e = [i[4] for i in batch]
    return (a, b, c, d, _F.one_hot(_torch.as_tensor(e), num_classes=max(num_categories, 105)).squeeze(1).float())


def _label_width(data_loader) -> int:
    """Width of the one-hot block the loader's collate function produces: max(number of classes, 105)."""
    fn = data_loader.collate_fn
    if isinstance(fn, partial) and fn.args:
        return max(int(fn.args[0]), 105)
    labels = data_loader.dataset.tensors[-1]
    return max(int(labels.max()) + 1, 105)


def make_dataloader_concat(rpkm, tnf, lengths, labels, batchsize: int = 256, destroy: bool = False, cuda: bool = False):
    """semisupervised_encode.py:111-148: the four tensors of ``make_dataloader`` + the integer class of every contig."""
    base = _encode.make_dataloader(rpkm, tnf, lengths, batchsize, destroy, cuda)
    labels_int = _np.unique(labels, return_inverse=True)[1]
    tensors = base.dataset.tensors
    dataset = _TensorDataset(*tensors, _torch.from_numpy(labels_int))
    prepared = getattr(base.dataset, "_vambhip_prepared", None)
    if prepared is not None:          # features normalised on the device: the resident matrix is shared, not copied
        dataset._vambhip_prepared = prepared
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=len(tensors[0]) > batchsize, shuffle=True,
                       num_workers=0, pin_memory=False, collate_fn=partial(collate_fn_concat, len(set(labels_int))))


def make_dataloader_labels(rpkm, tnf, lengths, labels, batchsize: int = 256, destroy: bool = False, cuda: bool = False):
    """semisupervised_encode.py:151-175 (the feature arrays only contribute their length and the argument checks)."""
    if batchsize < 1:
        raise ValueError(f"Batch size must be minimum 1, not {batchsize}")
    if len(rpkm) != len(tnf) or len(tnf) != len(lengths):
        raise ValueError("Lengths of abundance, TNF and lengths arrays must be the same")
    labels_int = _np.unique(labels, return_inverse=True)[1]
    dataset = _TensorDataset(_torch.from_numpy(labels_int))
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=len(rpkm) > batchsize, shuffle=True, num_workers=0,
                       pin_memory=False, collate_fn=partial(collate_fn_labels, len(set(labels_int))))


class _LabelledDeviceDataset:
    """A ``vh_dataset`` with labels: owns the handle unless it was borrowed from a device-prepared loader."""

    def __init__(self, lib, key, handle, owned: bool, keepalive=None):
        self._lib, self.key, self.handle, self._owned, self._keepalive = lib, key, handle, owned, keepalive

    def __del__(self):
        try:
            if self._owned and self.handle is not None:
                self._lib.vh_dataset_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


def _labels_i32(t, nlabels: int) -> _np.ndarray:
    arr = _np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, _torch.Tensor) else t).reshape(-1)
    if arr.size and (arr.min() < 0 or arr.max() >= nlabels):
        raise ValueError(f"labels must lie in [0, {nlabels}), got [{arr.min()}, {arr.max()}]")
    return arr.astype(_np.int32)


class _LabelledMixin:
    """What the two subclasses share: the native handle with a label block, label statistics, forward on explicit rows."""

    _KIND = 0
    _OPTIMIZER = VH_OPT_DADAPT_ADAM

    def _create_handle(self, cfg) -> ctypes.c_void_p:
        lab = _LabelsConfig(self._KIND, self.nlabels, self._OPTIMIZER, 1e-3)
        h = ctypes.c_void_p()
        _lib.check(self._lib.vh_vae_create_labelled(ctypes.byref(cfg), ctypes.byref(lab), ctypes.byref(h)))
        return h

    def attach_communicator(self, comm, syncbn: bool = True) -> None:
        """The label models train on one GPU: refused HERE, not after the dataset has been uploaded (VAE.attach_communicator
        would succeed and the first epoch would fail)."""
        if comm is not None:
            raise NotImplementedError("data-parallel training of VAELabels / VAEConcat is not wired into the host mirror; "
                                      "train them on one GPU")
        super().attach_communicator(None, syncbn)

    def _forward_rows(self, x: _np.ndarray, eps, masks):
        b = len(x)
        r = _np.empty((b, self._row_width()), _np.float32)
        mu = _np.empty((b, self.nlatent), _np.float32)
        e = None if eps is None else _encode._as_f32(eps)
        m = None if masks is None else _np.ascontiguousarray(
            _np.concatenate([_np.asarray(k, dtype=_np.uint8).reshape(-1) for k in masks]))
        _lib.check(self._lib.vh_vae_forward_rows(self._h, _lib.ptr(x), b, int(self.training), _lib.ptr(e), _lib.ptr(m),
                                                 _lib.ptr(r), _lib.ptr(mu)))
        return r, mu

    def _label_stats(self, n_epochs: int) -> _np.ndarray:
        out = _np.empty((n_epochs, 2), _np.float64)
        _lib.check(self._lib.vh_vae_label_stats(self._h, n_epochs, _lib.ptr(out)))
        return out

    def train_batch(self, rows, eps=None, masks=None):
        """One optimisation step on explicit dataset rows (parity tests): the five means of VAE.train_batch followed by
        (ce_labels, correct_labels)."""
        five = super().train_batch(rows, eps, masks)
        return five + tuple(self._label_stats(1)[0])

    def _run_epochs(self, data_loader, first_epoch: int, count: int, batchsteps):
        """`count` epochs from `first_epoch` (only the first may change the batch size); returns the loader, the [count][5]
        loss means and the [count][2] label statistics."""
        n_seq = self._ensure_dataset(data_loader)
        if n_seq < 2:
            raise ValueError(f"Cannot train on a dataset with fewer than 2 sequences, but got {n_seq} sequences.")
        self.train()
        if first_epoch in batchsteps:
            new_bs = data_loader.batch_size * 2
            data_loader = _DataLoader(dataset=data_loader.dataset, batch_size=new_bs, shuffle=True, drop_last=n_seq > new_bs,
                                      num_workers=0, pin_memory=False, collate_fn=data_loader.collate_fn)
        bs = data_loader.batch_size
        n_batches, batch = (n_seq // bs, bs) if n_seq > bs else (1, n_seq)
        means = (ctypes.c_double * (5 * count))()
        _lib.check(self._lib.vh_vae_train_epochs(self._h, count, n_batches, batch, 0, means))
        return data_loader, _np.array(means, _np.float64).reshape(count, 5), self._label_stats(count), n_batches


# ---- VAELabels (semisupervised_encode.py:189-436) ---------------------------------------------------------------------
class VAELabels(_LabelledMixin, _encode.VAE):
    """Variational autoencoder that encodes only the one-hot labels, subclass of VAE.
        nlabels: width of the one-hot block; the other arguments as VAE."""

    _KIND = VH_VAE_LABELS
    _OPTIMIZER = VH_OPT_ADAM

    def __init__(self, nlabels: int, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha: Optional[float] = None, beta: float = 200, dropout: Optional[float] = 0.2, cuda: bool = False,
                 _seed: int = 0):
        self.nlabels = nlabels
        super().__init__(nlabels - 104, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout,
                         cuda=cuda, seed=_seed)
        self.nlabels = nlabels

    def _row_width(self) -> int:
        return self.nlabels

    def forward(self, labels, _eps=None, _masks=None):
        x = _encode._as_f32(labels)
        if x.ndim != 2 or x.shape[1] != self.nlabels:
            raise ValueError(f"expected one-hot labels [B, {self.nlabels}]")
        r, mu = self._forward_rows(x, _eps, _masks)
        mu = _torch.from_numpy(mu)
        return _torch.from_numpy(r), mu, _torch.zeros(mu.size())

    __call__ = forward

    def calc_loss(self, labels_in, labels_out, mu, logsigma):
        """semisupervised_encode.py:248-257 on host tensors (training computes the same on the device)."""
        t = lambda x: x if isinstance(x, _torch.Tensor) else _torch.as_tensor(x)  # noqa: E731
        labels_in, labels_out, mu, logsigma = t(labels_in), t(labels_out), t(mu), t(logsigma)
        _, labels_in_indices = labels_in.max(dim=1)
        ce_labels = _torch.nn.CrossEntropyLoss()(labels_out, labels_in_indices)
        kld = -0.5 * (1 + logsigma - mu.pow(2) - logsigma.exp()).sum(dim=1).mean()
        loss = ce_labels * 1.0 + kld * (1 / (self.nlatent * self.beta))
        _, labels_out_indices = labels_out.max(dim=1)
        return loss, ce_labels, kld, _torch.sum(labels_out_indices == labels_in_indices)

    def _ensure_dataset(self, data_loader) -> int:
        holder = data_loader.dataset
        tensors = holder.tensors
        if len(tensors) != 1:
            raise ValueError("expected a DataLoader made by make_dataloader_labels (1 tensor)")
        if _label_width(data_loader) != self.nlabels:
            raise ValueError(f"the loader one-hots to {_label_width(data_loader)} columns, the model has {self.nlabels}")
        lab = tensors[0]
        key = (lab.data_ptr(), tuple(lab.shape), lab._version)
        same = self._dataset_ref is not None and self._dataset_ref() is holder
        if not same or key != self._dataset_key:
            cached = getattr(holder, "_vambhip_device_labels", None)
            if cached is None or cached.key != key:
                arr = _labels_i32(lab, self.nlabels)
                h = ctypes.c_void_p()
                _lib.check(self._lib.vh_dataset_create_labels(_lib.ptr(arr), len(arr), self.nlabels, ctypes.byref(h)))
                cached = _LabelledDeviceDataset(self._lib, key, h, owned=True)
                holder._vambhip_device_labels = cached
            _lib.check(self._lib.vh_vae_use_dataset(self._h, cached.handle))
            self._device_dataset, self._dataset_key, self._dataset_ref = cached, key, weakref.ref(holder)
            self._n_rows = len(lab)
        return self._n_rows

    def trainepoch(self, data_loader, epoch, optimizer, batchsteps):
        """One epoch (semisupervised_encode.py:259-314); `optimizer` is accepted for signature compatibility (the Adam state
        lives in the native handle)."""
        data_loader, means, stats, n_batches = self._run_epochs(data_loader, epoch, 1, batchsteps)
        self._log_epochs(epoch, means, stats, n_batches, data_loader.batch_size)
        return data_loader

    def _log_epochs(self, first_epoch, means, stats, n_batches, bs):
        for e in range(len(means)):
            # (the reference divides the number of correct labels by len(data_loader) * 256 whatever the batch size, :309)
            logger.info("\tEpoch: {}\tLoss: {:.6f}\tCE_labels: {:.7f}\tKLD: {:.4f}\taccuracy: {:.4f}\tBatchsize: {}".format(
                first_epoch + e + 1, means[e][0], stats[e][0], means[e][4] * (self.nlatent * self.beta),
                stats[e][1] / (n_batches * 256), bs))
        self.last_epoch_losses = dict(loss=means[-1][0], ce_labels=stats[-1][0], kld=means[-1][4] * (self.nlatent * self.beta),
                                      correct_labels=stats[-1][1], batchsize=bs)

    def trainmodel(self, dataloader, nepochs: int = 500, lrate: float = 1e-3, batchsteps: Optional[list[int]] = [25, 75, 150, 300],
                   modelfile=None):
        """semisupervised_encode.py:362-436.  Output: None"""
        if lrate < 0:
            raise ValueError(f"Learning rate must be positive, not {lrate}")
        if nepochs < 1:
            raise ValueError(f"Minimum 1 epoch, not {nepochs}")
        if batchsteps is None:
            batchsteps_set: set[int] = set()
        else:
            batchsteps = list(batchsteps)
            if not all(isinstance(i, int) for i in batchsteps):
                raise ValueError("All elements of batchsteps must be integers")
            if max(batchsteps, default=0) >= nepochs:
                raise ValueError("Max batchsteps must not equal or exceed nepochs")
            batchsteps_set = set(batchsteps)
        logger.info("\tNetwork properties:")
        logger.info(f"\t    CUDA: {self.usecuda}")
        logger.info(f"\t    Alpha: {self.alpha}")
        logger.info(f"\t    Beta: {self.beta}")
        logger.info(f"\t    Dropout: {self.dropout}")
        logger.info(f"\t    N hidden: {', '.join(map(str, self.nhiddens))}")
        logger.info(f"\t    N latent: {self.nlatent}")
        logger.info("\tTraining properties:")
        logger.info(f"\t    N epochs: {nepochs}")
        logger.info(f"\t    Starting batch size: {dataloader.batch_size}")
        steps = ", ".join(map(str, sorted(batchsteps_set))) if batchsteps_set else "None"
        logger.info(f"\t    Batchsteps: {steps}")
        logger.info(f"\t    Learning rate: {lrate}")
        logger.info(f"\t    N labels: {dataloader.dataset.tensors[0].shape}")
        # `optimizer = Adam(self.parameters(), lr=lrate)` (:405): a fresh state per call
        _lib.check(self._lib.vh_vae_set_optimizer(self._h, VH_OPT_ADAM, float(lrate)))
        _lib.check(self._lib.vh_vae_reset_optimizer(self._h))
        epoch = 0
        while epoch < nepochs:   # the epochs between two batch-size changes go out as ONE library call
            nxt = min([b for b in batchsteps_set if b > epoch] + [nepochs])
            dataloader, means, stats, n_batches = self._run_epochs(dataloader, epoch, nxt - epoch, batchsteps_set)
            self._log_epochs(epoch, means, stats, n_batches, dataloader.batch_size)
            epoch = nxt
        self.eval()
        if modelfile is not None:
            try:
                self.save(modelfile)
            except Exception:
                pass
        return None


# ---- VAEConcat (semisupervised_encode.py:438-698) ---------------------------------------------------------------------
class VAEConcat(_LabelledMixin, _encode.VAE):
    """Variational autoencoder that uses TNFs, abundances and labels as concatenated input, subclass of VAE.
        nsamples: Number of samples in abundance matrix;  nlabels: width of the one-hot block;  the rest as VAE."""

    _KIND = VH_VAE_CONCAT

    def __init__(self, nsamples: int, nlabels: int, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha: Optional[float] = None, beta: float = 200.0, dropout: Optional[float] = 0.2, cuda: bool = False,
                 _seed: int = 0):
        if nsamples < 1:
            raise ValueError(f"nsamples must be > 0, not {nsamples}")
        self.nlabels = nlabels
        self._native_nsamples = nsamples
        # as the reference: the defaults of alpha / nhiddens / dropout see nsamples + nlabels (:466-474)
        super().__init__(nsamples + nlabels, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout,
                         cuda=cuda, seed=_seed)
        self.nsamples = nsamples
        self.nlabels = nlabels

    def _row_width(self) -> int:
        return self._native_nsamples + NTNF + 1 + self.nlabels

    def forward(self, depths, tnf, abundance, labels, _eps=None, _masks=None):
        d, t, a, l = (_encode._as_f32(x) for x in (depths, tnf, abundance, labels))
        if d.ndim != 2 or d.shape[1] != self.nsamples or t.shape != (len(d), NTNF) or a.shape != (len(d), 1) \
                or l.shape != (len(d), self.nlabels):
            raise ValueError("expected depths [B, nsamples], tnf [B, 103], abundance [B, 1], labels [B, nlabels]")
        r, mu = self._forward_rows(_np.ascontiguousarray(_np.concatenate((d, t, a, l), axis=1)), _eps, _masks)
        s = self.nsamples
        f = lambda x: _torch.from_numpy(_np.ascontiguousarray(x))  # noqa: E731
        mu = _torch.from_numpy(mu)
        return (f(r[:, :s]), f(r[:, s:s + NTNF]), f(r[:, s + NTNF:s + NTNF + 1]), f(r[:, s + NTNF + 1:]), mu,
                _torch.zeros(mu.size()))

    __call__ = forward

    def calc_loss(self, depths_in, depths_out, tnf_in, tnf_out, abundance_in, abundance_out, labels_in, labels_out, mu,
                  logsigma, weights):
        """semisupervised_encode.py:515-569 on host tensors (note the [B] x [B,1] broadcast of `weights`, as in VAE)."""
        t = lambda x: x if isinstance(x, _torch.Tensor) else _torch.as_tensor(x)  # noqa: E731
        depths_in, depths_out, tnf_in, tnf_out = t(depths_in), t(depths_out), t(tnf_in), t(tnf_out)
        abundance_in, abundance_out, labels_in, labels_out = t(abundance_in), t(abundance_out), t(labels_in), t(labels_out)
        mu, weights = t(mu), t(weights)
        ab_sse = (abundance_out - abundance_in).pow(2).sum(dim=1)
        ce = -((depths_out + 1e-9).log() * depths_in).sum(dim=1)
        sse = (tnf_out - tnf_in).pow(2).sum(dim=1)
        kld = 0.5 * (mu.pow(2)).sum(dim=1)
        if self.nsamples == 1:
            ce_weight = 0.0
        else:
            ce_weight = ((1 - self.alpha) * (self.nsamples - 1)) / (self.nsamples * _log(self.nsamples))
        ab_sse_weight = (1 - self.alpha) * (1 / self.nsamples)
        sse_weight = self.alpha / self.ntnf
        kld_weight = 1 / (self.nlatent * self.beta)
        _, labels_in_indices = labels_in.max(dim=1)
        ce_labels = _torch.nn.CrossEntropyLoss()(labels_out, labels_in_indices)
        reconstruction_loss = ce * ce_weight + ab_sse * ab_sse_weight + sse * sse_weight + ce_labels * 1.0
        loss = (reconstruction_loss + kld * kld_weight) * weights
        _, labels_out_indices = labels_out.max(dim=1)
        return loss, ce, sse, ce_labels, kld, _torch.sum(labels_out_indices == labels_in_indices)

    def _ensure_dataset(self, data_loader) -> int:
        holder = data_loader.dataset
        tensors = holder.tensors
        if len(tensors) != 5:
            raise ValueError("expected a DataLoader made by make_dataloader_concat (5 tensors)")
        if _label_width(data_loader) != self.nlabels:
            raise ValueError(f"the loader one-hots to {_label_width(data_loader)} columns, the model has {self.nlabels}")
        lab = tensors[4]
        n = len(lab)
        prepared = getattr(holder, "_vambhip_prepared", None)
        lab_key = (lab.data_ptr(), tuple(lab.shape), lab._version)
        if prepared is not None:
            key = ("prepared", id(prepared), lab_key)
        else:
            key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in tensors)
        same = self._dataset_ref is not None and self._dataset_ref() is holder
        if not same or key != self._dataset_key:
            cached = getattr(holder, "_vambhip_device_labels", None)
            if cached is None or cached.key != key:
                arr = _labels_i32(lab, self.nlabels)
                if prepared is not None:
                    if prepared.nsamples != self.nsamples:
                        raise ValueError("dataset tensors do not match this VAE (nsamples / 103 TNF / 1 / 1 columns)")
                    handle, owned = prepared.handle, False
                else:
                    d, t, a, w = (_encode._as_f32(x) for x in tensors[:4])
                    if d.shape != (n, self.nsamples) or t.shape != (n, NTNF) or a.shape != (n, 1) or w.shape != (n, 1):
                        raise ValueError("dataset tensors do not match this VAE (nsamples / 103 TNF / 1 / 1 columns)")
                    handle = ctypes.c_void_p()
                    _lib.check(self._lib.vh_dataset_create(_lib.ptr(d), _lib.ptr(t), _lib.ptr(a), _lib.ptr(w), n,
                                                           self.nsamples, ctypes.byref(handle)))
                    owned = True
                _lib.check(self._lib.vh_dataset_set_labels(handle, _lib.ptr(arr), n, self.nlabels))
                cached = _LabelledDeviceDataset(self._lib, key, handle, owned, keepalive=prepared)
                holder._vambhip_device_labels = cached
            _lib.check(self._lib.vh_vae_use_dataset(self._h, cached.handle))
            self._device_dataset, self._dataset_key, self._dataset_ref = cached, key, weakref.ref(holder)
            self._n_rows = n
        return self._n_rows

    def trainepoch(self, data_loader, epoch, optimizer, batchsteps):
        """One epoch (semisupervised_encode.py:571-649)."""
        data_loader, means, stats, n_batches = self._run_epochs(data_loader, epoch, 1, batchsteps)
        self._log_epochs(epoch, means, stats, n_batches, data_loader.batch_size)
        self.eval()
        return data_loader

    def _train_segment(self, data_loader, first_epoch: int, count: int, batchsteps):
        """The inherited VAE.trainmodel (the reference's VAEConcat has no trainmodel of its own) drives this."""
        data_loader, means, stats, n_batches = self._run_epochs(data_loader, first_epoch, count, batchsteps)
        self._log_epochs(first_epoch, means, stats, n_batches, data_loader.batch_size)
        self.eval()
        return data_loader

    def _log_epochs(self, first_epoch, means, stats, n_batches, bs):
        ce_w = 0.0 if self.nsamples == 1 else ((1 - self.alpha) * (self.nsamples - 1)) / (self.nsamples * _log(self.nsamples))
        sse_w, kld_w = self.alpha / self.ntnf, 1 / (self.nlatent * self.beta)
        for e in range(len(means)):
            loss, ab, ce, sse, kld = means[e]   # weighted means (VAE.calc_loss order); the reference logs the raw ones here
            logger.info("\tEpoch: {}\tLoss: {:.6f}\tCE: {:.7f}\tSSE: {:.6f}\tCE_labels: {:.7f}\tKLD: {:.4f}\taccuracy: {:.4f}"
                        "\tBatchsize: {}".format(first_epoch + e + 1, loss, ce / ce_w if ce_w else 0.0, sse / sse_w,
                                                 stats[e][0], kld / kld_w, stats[e][1] / n_batches, bs))
        self.last_epoch_losses = dict(loss=means[-1][0], ab=means[-1][1], ce=means[-1][2], sse=means[-1][3], kld=means[-1][4],
                                      ce_labels=stats[-1][0], correct_labels=stats[-1][1], batchsize=bs)
