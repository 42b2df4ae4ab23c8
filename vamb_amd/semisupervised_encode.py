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
        lab = _LabelsConfig(self._KIND, self._NL, self._OPTIMIZER, 1e-3)
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
        self._NL = nlabels   # width of the label block in the network (the HLoss subclasses overwrite `nlabels`)
        super().__init__(nlabels - 104, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout,
                         cuda=cuda, seed=_seed)
        self.nlabels = nlabels

    def _row_width(self) -> int:
        return self._NL

    def forward(self, labels, _eps=None, _masks=None):
        x = _encode._as_f32(labels)
        if x.ndim != 2 or x.shape[1] != self._NL:
            raise ValueError(f"expected one-hot labels [B, {self._NL}]")
        r, mu = self._forward_rows(x, _eps, _masks)
        mu = _torch.from_numpy(mu)
        # `_decode` narrows to self.nlabels columns (:236): all of them here, the leaves in VAELabelsHLoss
        return _torch.from_numpy(_np.ascontiguousarray(r[:, :self.nlabels])), mu, _torch.zeros(mu.size())

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
        if _label_width(data_loader) != self._NL:
            raise ValueError(f"the loader one-hots to {_label_width(data_loader)} columns, the model has {self._NL}")
        lab = tensors[0]
        key = (lab.data_ptr(), tuple(lab.shape), lab._version)
        same = self._dataset_ref is not None and self._dataset_ref() is holder
        if not same or key != self._dataset_key:
            cached = getattr(holder, "_vambhip_device_labels", None)
            if cached is None or cached.key != key:
                arr = _labels_i32(lab, self._NL)
                h = ctypes.c_void_p()
                _lib.check(self._lib.vh_dataset_create_labels(_lib.ptr(arr), len(arr), self._NL, ctypes.byref(h)))
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
        return self._trainmodel(dataloader, nepochs, lrate, batchsteps, modelfile, VH_OPT_ADAM)

    def _trainmodel(self, dataloader, nepochs, lrate, batchsteps, modelfile, optimizer):
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
        # `optimizer = Adam(self.parameters(), lr=lrate)` (:405) -- DAdaptAdam(lr=1) in VAELabelsHLoss: a fresh state per call
        _lib.check(self._lib.vh_vae_set_optimizer(self._h, optimizer, float(lrate)))
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
        self._NL = nlabels   # width of the label block in the network (the HLoss subclasses overwrite `nlabels`)
        self._native_nsamples = nsamples
        # as the reference: the defaults of alpha / nhiddens / dropout see nsamples + nlabels (:466-474)
        super().__init__(nsamples + nlabels, nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout,
                         cuda=cuda, seed=_seed)
        self.nsamples = nsamples
        self.nlabels = nlabels

    def _row_width(self) -> int:
        return self._native_nsamples + NTNF + 1 + self._NL

    def forward(self, depths, tnf, abundance, labels, _eps=None, _masks=None):
        d, t, a, l = (_encode._as_f32(x) for x in (depths, tnf, abundance, labels))
        if d.ndim != 2 or d.shape[1] != self.nsamples or t.shape != (len(d), NTNF) or a.shape != (len(d), 1) \
                or l.shape != (len(d), self._NL):
            raise ValueError("expected depths [B, nsamples], tnf [B, 103], abundance [B, 1], labels [B, nlabels]")
        r, mu = self._forward_rows(_np.ascontiguousarray(_np.concatenate((d, t, a, l), axis=1)), _eps, _masks)
        s = self.nsamples
        f = lambda x: _torch.from_numpy(_np.ascontiguousarray(x))  # noqa: E731
        mu = _torch.from_numpy(mu)
        return (f(r[:, :s]), f(r[:, s:s + NTNF]), f(r[:, s + NTNF:s + NTNF + 1]),
                f(r[:, s + NTNF + 1:s + NTNF + 1 + self.nlabels]), mu, _torch.zeros(mu.size()))

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
        if _label_width(data_loader) != self._NL:
            raise ValueError(f"the loader one-hots to {_label_width(data_loader)} columns, the model has {self._NL}")
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
                arr = _labels_i32(lab, self._NL)
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
                _lib.check(self._lib.vh_dataset_set_labels(handle, _lib.ptr(arr), n, self._NL))
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


# ---- the joint trainer (semisupervised_encode.py:50-87, 178-186, 700-1145) ---------------------------------------------
def collate_fn_semisupervised(num_categories: int, batch):
    cols = [[i[k] for i in batch] for k in range(10)]
    hot = lambda v: _F.one_hot(_torch.as_tensor(v), num_classes=max(num_categories, 105)).squeeze(1).float()  # noqa: E731
    st = _torch.stack
    return (st(cols[0]), st(cols[1]), st(cols[2]), st(cols[3]), hot(cols[4]), st(cols[5]), st(cols[6]), st(cols[7]), st(cols[8]),
            hot(cols[9]))


def kld_gauss(p_mu, p_logstd, q_mu, q_logstd):
    """semisupervised_encode.py:79-86 (host tensors)."""
    loss = q_logstd - p_logstd + (p_logstd.exp().pow(2) + (p_mu - q_mu).pow(2)) / (2 * q_logstd.exp().pow(2)) - 0.5
    return loss.mean()


def permute_indices(n_current: int, n_total: int, seed: int):
    """semisupervised_encode.py:178-186."""
    rng = _np.random.default_rng(seed)
    x = _np.arange(n_current)
    to_add = int(n_total / n_current)
    to_concatenate = [rng.permutation(x)]
    for _ in range(to_add):
        to_concatenate.append(rng.permutation(x))
    return _np.concatenate(to_concatenate)[:n_total]


VAEVAE_METRICS = ["loss_vamb", "ab_vamb", "ce_vamb", "sse_vamb", "kld_vamb", "loss_labels", "ce_labels_labels", "kld_labels",
                  "correct_labels_labels", "loss_joint", "ce_joint", "sse_joint", "ce_labels_joint", "kld_vamb_joint",
                  "kld_labels_joint", "correct_labels_joint", "loss"]   # semisupervised_encode.py:830-848


class _JointDeviceDatasets:
    """The three row-aligned vh_datasets of one semisupervised loader (owned)."""

    def __init__(self, lib, key, unsup, unsup_labels, sup):
        self._lib, self.key, self.handles = lib, key, [unsup, unsup_labels, sup]

    def __del__(self):
        try:
            for h in self.handles:
                if h is not None:
                    self._lib.vh_dataset_destroy(h)
            self.handles = []
        except Exception:
            pass


class VAEVAE(object):
    """Bi-modal variational autoencoder that uses TNFs, abundances and one-hot labels: three encoders (VAMB, labels,
    concatenated VAMB + labels) and two decoders (VAMB and labels) -- semisupervised_encode.py:700-1145.

    The three networks are ``vamb_amd`` models (``VAEVamb``, ``VAELabels``, ``VAEJoint``: usable on their own for encoding and
    state dicts); a training step -- seven passes through them, the sum of three losses, one Adam update -- runs inside
    libvambhip (``vh_vaevae_*``, csrc/vaevae.hpp).  (The reference's own base class cannot train: its ``trainepoch`` reads
    ``self.usecuda`` / ``self.alpha``, which only the subclass ``vamb.taxvamb_encode.VAEVAEHLoss`` defines, :917, :789.  This one
    runs the computation that code spells out, with the one-hot cross-entropy of its ``calc_loss_joint``.)"""

    def __init__(self, nsamples: int, nlabels: int, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha: Optional[float] = None, beta: float = 200.0, dropout: Optional[float] = 0.2, cuda: bool = False):
        N_l = max(nlabels, 105)
        self.usecuda = cuda
        kw = dict(nhiddens=nhiddens, nlatent=nlatent, alpha=alpha, beta=beta, dropout=dropout, cuda=cuda)
        # (distinct seeds: the three networks draw independent dropout / noise / initialisation streams)
        self.VAEVamb = _encode.VAE(nsamples, seed=0, **kw)
        self.VAELabels = VAELabels(N_l, _seed=1, **kw)
        self.VAEJoint = VAEConcat(nsamples, N_l, _seed=2, **kw)

    # ---- native trainer -------------------------------------------------------------------------------------------------
    def _networks(self):
        return (self.VAEVamb, self.VAELabels, self.VAEJoint)

    def _trainer(self):
        t = getattr(self, "_vv", None)
        if t is None:
            lib = _lib.load()
            for net in self._networks():   # the joint step is an fp32 step
                if net.compute_dtype != "fp32":
                    _lib.check(lib.vh_vae_set_precision(net._h, 0))
                    net.compute_dtype = "fp32"
            t = ctypes.c_void_p()
            _lib.check(lib.vh_vaevae_create(self.VAEVamb._h, self.VAELabels._h, self.VAEJoint._h, ctypes.byref(t)))
            self._vv, self._vv_lib = t, lib
            self._vv_key = self._vv_ref = None
        return t

    def __del__(self):
        try:
            t = getattr(self, "_vv", None)
            if t is not None and t.value:
                self._vv_lib.vh_vaevae_destroy(t)
                self._vv = None
        except Exception:
            pass

    def _set_adam(self, lrate: float, reset: bool) -> None:
        lib = _lib.load()
        for net in self._networks():
            _lib.check(lib.vh_vae_set_optimizer(net._h, VH_OPT_ADAM, float(lrate)))
            if reset:
                _lib.check(lib.vh_vae_reset_optimizer(net._h))
        self._adam_lrate = float(lrate)

    def _require_adam(self, optimizer=None) -> None:
        """trainepoch / train_batch called before trainmodel: the reference takes the optimiser as an argument (its learning rate
        is what counts: the Adam state lives in the native handles), so use it -- or say clearly what is missing."""
        groups = getattr(optimizer, "param_groups", None)
        current = getattr(self, "_adam_lrate", None)
        if current is not None:
            # an optimiser passed to trainepoch steps with ITS learning rate (semisupervised_encode.py:829-997); the moments
            # the native handles hold are kept
            if groups and float(groups[0]["lr"]) != current:
                self._set_adam(float(groups[0]["lr"]), reset=False)
            return
        if groups:
            self._set_adam(float(groups[0]["lr"]), reset=True)
            return
        raise ValueError("no Adam learning rate set: call trainmodel(...), or pass the torch optimizer whose `lr` should be used "
                         "to trainepoch (the optimiser state itself lives in the native handles)")

    def _ensure_dataset(self, data_loader) -> int:
        """Upload the ten tensors of the semisupervised loader (taxvamb_encode.py:213-224) as three resident datasets."""
        holder = data_loader.dataset
        tensors = holder.tensors
        if len(tensors) != 10:
            raise ValueError("expected a DataLoader made by make_dataloader_semisupervised_hloss (10 tensors)")
        NL = self.VAELabels._NL
        if _label_width(data_loader) != NL:
            raise ValueError(f"the loader one-hots to {_label_width(data_loader)} columns, the model has {NL}")
        t = self._trainer()
        lib = self._vv_lib
        key = tuple((x.data_ptr(), tuple(x.shape), x._version) for x in tensors)
        same = self._vv_ref is not None and self._vv_ref() is holder
        if not same or key != self._vv_key:
            cached = getattr(holder, "_vambhip_joint_datasets", None)
            if cached is None or cached.key != key:
                n, S = len(tensors[0]), self.VAEVamb.nsamples
                handles = []
                for lo in (0, 5):
                    d, tn, a, w = (_encode._as_f32(x) for x in tensors[lo:lo + 4])
                    if d.shape != (n, S) or tn.shape != (n, NTNF) or a.shape != (n, 1) or w.shape != (n, 1):
                        raise ValueError("dataset tensors do not match this VAE (nsamples / 103 TNF / 1 / 1 columns)")
                    h = ctypes.c_void_p()
                    _lib.check(lib.vh_dataset_create(_lib.ptr(d), _lib.ptr(tn), _lib.ptr(a), _lib.ptr(w), n, S, ctypes.byref(h)))
                    handles.append(h)
                lab_u, lab_s = _labels_i32(tensors[4], NL), _labels_i32(tensors[9], NL)
                if len(lab_u) != n or len(lab_s) != n:
                    raise ValueError("the ten tensors must have the same number of rows")
                hl = ctypes.c_void_p()
                _lib.check(lib.vh_dataset_create_labels(_lib.ptr(lab_u), n, NL, ctypes.byref(hl)))
                _lib.check(lib.vh_dataset_set_labels(handles[1], _lib.ptr(lab_s), n, NL))
                cached = _JointDeviceDatasets(lib, key, handles[0], hl, handles[1])
                holder._vambhip_joint_datasets = cached
            self._vv_datasets, self._vv_key, self._vv_ref = cached, key, weakref.ref(holder)
            self._vv_rows = len(tensors[0])
        # (re)attach on every call: the three networks can be used on their own in between (VAEJoint.encode(dataloader_joint)
        # attaches ITS loader's dataset to the same native handle) -- and forget what they had attached, for the same reason
        _lib.check(lib.vh_vaevae_set_datasets(t, *self._vv_datasets.handles))
        for net in self._networks():
            net._dataset_key = net._dataset_ref = None
        return self._vv_rows

    def train_batch(self, rows, eps=None, masks=None):
        """One optimisation step on explicit rows of the attached loader (parity tests).  eps: per pass (joint, vamb_x, labels_x,
        vamb_u, vamb_s, labels_u, labels_s) a [B, nlatent] array; masks: per pass the list of its dropout keep-masks.  Returns
        the 17 metrics (VAEVAE_METRICS order)."""
        rows = _np.ascontiguousarray(rows, dtype=_np.int64)
        e = None if eps is None else _np.ascontiguousarray(_np.stack([_encode._as_f32(x) for x in eps]))
        m = None if masks is None else _np.ascontiguousarray(
            _np.concatenate([_np.asarray(k, dtype=_np.uint8).reshape(-1) for p in masks for k in p]))
        out = (ctypes.c_double * 17)()
        t = self._trainer()   # (creates the native trainer -- and self._vv_lib -- on first use)
        self._require_adam()
        _lib.check(self._vv_lib.vh_vaevae_train_step(t, _lib.ptr(rows), len(rows), _lib.ptr(e), _lib.ptr(m), out))
        return list(out)

    def get_grad(self, network: str, name: str) -> _np.ndarray:
        net = dict(VAEVamb=0, VAELabels=1, VAEJoint=2)[network]
        n = _lib._i64(0)
        t = self._trainer()
        _lib.check(self._vv_lib.vh_vae_param_size(self._networks()[net]._h, name.encode(), ctypes.byref(n)))
        out = _np.empty(n.value, _np.float32)
        _lib.check(self._vv_lib.vh_vaevae_get_grad(t, net, name.encode(), _lib.ptr(out), n.value))
        return out

    # ---- reference interface ----------------------------------------------------------------------------------------------
    def calc_loss_joint(self, depths_in, depths_out, tnf_in, tnf_out, abundance_in, abundance_out, labels_in, labels_out, mu_sup,
                        logsigma_sup, mu_vamb_unsup, logsigma_vamb_unsup, mu_labels_unsup, logsigma_labels_unsup, weights):
        """semisupervised_encode.py:762-827 on host tensors (training computes the same on the device)."""
        v = self.VAEVamb
        ab_sse = (abundance_out - abundance_in).pow(2).sum(dim=1)
        ce = -((depths_out + 1e-9).log() * depths_in).sum(dim=1)
        sse = (tnf_out - tnf_in).pow(2).sum(dim=1)
        ce_weight = 0.0 if v.nsamples == 1 else ((1 - v.alpha) * (v.nsamples - 1)) / (v.nsamples * _log(v.nsamples))
        ab_sse_weight = (1 - v.alpha) * (1 / v.nsamples)
        sse_weight = v.alpha / v.ntnf
        ce_labels, correct = self._label_loss(labels_out, labels_in)
        reconstruction_loss = ce * ce_weight + ab_sse * ab_sse_weight + sse * sse_weight + ce_labels * 1.0
        kld_vamb = kld_gauss(mu_sup, logsigma_sup, mu_vamb_unsup, logsigma_vamb_unsup)
        kld_labels = kld_gauss(mu_sup, logsigma_sup, mu_labels_unsup, logsigma_labels_unsup)
        kld_loss = (kld_vamb + kld_labels) * (1 / (v.nlatent * v.beta))
        loss = (reconstruction_loss + kld_loss) * weights
        return loss.mean(), ce.mean(), sse.mean(), ce_labels.mean(), kld_vamb.mean(), kld_labels.mean(), correct

    def _label_loss(self, labels_out, labels_in):
        _, idx = labels_in.max(dim=1)
        _, out_idx = labels_out.max(dim=1)
        return _torch.nn.CrossEntropyLoss()(labels_out, idx), _torch.sum(out_idx == idx)

    def trainepoch(self, data_loader, epoch, optimizer, batchsteps):
        """One epoch (semisupervised_encode.py:829-1008); `optimizer` is accepted for signature compatibility (the Adam state
        lives in the three native handles).  Rows are taken in the loader's order: sequential until the first batch-size
        doubling replaces the loader by a shuffling one (:852-862)."""
        n = self._ensure_dataset(data_loader)
        self._require_adam(optimizer)
        if epoch in batchsteps:
            new_bs = data_loader.batch_size * 2
            data_loader = _DataLoader(dataset=data_loader.dataset, batch_size=new_bs, shuffle=True, drop_last=n > new_bs,
                                      num_workers=0, pin_memory=False, collate_fn=data_loader.collate_fn)
        bs = data_loader.batch_size
        order = _np.fromiter(iter(data_loader.sampler), dtype=_np.int64, count=n)
        if data_loader.drop_last:
            n_batches, batch = n // bs, bs
        else:
            if n > bs:
                raise ValueError("a loader that keeps a ragged last batch is not supported: use drop_last (the reference's "
                                 "loaders drop it whenever the dataset is larger than the batch)")
            n_batches, batch = 1, n
        if n_batches < 1 or batch < 2:
            raise ValueError(f"cannot train on {n} sequences with batch size {bs}")
        rows = _np.ascontiguousarray(order[:n_batches * batch])
        out = (ctypes.c_double * 17)()
        _lib.check(self._vv_lib.vh_vaevae_train_epoch(self._trainer(), _lib.ptr(rows), n_batches, batch, out))
        self.last_epoch_metrics = dict(zip(VAEVAE_METRICS, out))
        logger.info(f"\t\tEpoch: {epoch}  " + "  ".join(k + f": {v:.5e}" for k, v in self.last_epoch_metrics.items()))
        return data_loader

    def trainmodel(self, dataloader, nepochs: int = 500, lrate: float = 1e-3, batchsteps: list[int] = [25, 75, 150, 300],
                   modelfile=None):
        """Train the three networks jointly (semisupervised_encode.py:1010-1084).  Output: None"""
        if lrate < 0:
            raise ValueError("Learning rate must be positive, not {}".format(lrate))
        if nepochs < 1:
            raise ValueError("Minimum 1 epoch, not {}".format(nepochs))
        if batchsteps is None:
            batchsteps_set = set()
        else:
            batchsteps = list(batchsteps)
            if not all(isinstance(i, int) for i in batchsteps):
                raise ValueError("All elements of batchsteps must be integers")
            if max(batchsteps, default=0) >= nepochs:
                raise ValueError("Max batchsteps must not equal or exceed nepochs")
            batchsteps_set = set(batchsteps)
        ncontigs, nsamples = dataloader.dataset.tensors[0].shape
        v = self.VAEVamb
        self._trainer()
        self._set_adam(lrate, reset=True)   # `optimizer = Adam(all parameters, lr=lrate)`: a fresh state per call (:1048)
        logger.info("\tNetwork properties:")
        logger.info(f"\t    CUDA: {v.usecuda}")
        logger.info(f"\t    Alpha: {v.alpha}")
        logger.info(f"\t    Beta: {v.beta}")
        logger.info(f"\t    Dropout: {v.dropout}")
        logger.info(f"\t    N hidden: {', '.join(map(str, v.nhiddens))}")
        logger.info(f"\t    N latent: {v.nlatent}")
        logger.info("\tTraining properties:")
        logger.info(f"\t    N epochs: {nepochs}")
        logger.info(f"\t    Starting batch size: {dataloader.batch_size}")
        batchsteps_string = ", ".join(map(str, sorted(batchsteps_set))) if batchsteps_set else "None"
        logger.info(f"\t    Batchsteps: {batchsteps_string}")
        logger.info(f"\t    Learning rate: {lrate}")
        logger.info(f"\t    N sequences: {ncontigs}")
        logger.info(f"\t    N samples: {nsamples}")
        for epoch in range(nepochs):
            dataloader = self.trainepoch(dataloader, epoch, None, batchsteps_set)
        if modelfile is not None:
            try:
                self.save(modelfile)
            except Exception:
                pass
        return None

    def save(self, filehandle):
        """semisupervised_encode.py:1086-1104."""
        v = self.VAEVamb
        state = {"nsamples": v.nsamples, "nlabels": self.VAELabels.nlabels, "alpha": v.alpha, "beta": v.beta, "dropout": v.dropout,
                 "nhiddens": v.nhiddens, "nlatent": v.nlatent, "state_VAEVamb": self.VAEVamb.state_dict(),
                 "state_VAELabels": self.VAELabels.state_dict(), "state_VAEJoint": self.VAEJoint.state_dict()}
        _torch.save(state, filehandle)

    @classmethod
    def load(cls, path, cuda=False, evaluate=True):
        """semisupervised_encode.py:1106-1145."""
        d = _torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
        vae = cls(d["nsamples"], d["nlabels"], d["nhiddens"], d["nlatent"], d["alpha"], d["beta"], d["dropout"], cuda)
        vae.VAEVamb.load_state_dict(d["state_VAEVamb"])
        vae.VAELabels.load_state_dict(d["state_VAELabels"])
        vae.VAEJoint.load_state_dict(d["state_VAEJoint"])
        if evaluate:
            for net in vae._networks():
                net.eval()
        return vae
