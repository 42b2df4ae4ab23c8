"""VAE training / encoding on MI355X -- drop-in for ``vamb.encode``.

``make_dataloader``, ``set_batchsize`` and ``VAE`` keep the reference's signatures, argument meaning
and ``ValueError`` behaviour (``/root/reference/vamb/encode.py:33-610``) so that
``vamb.__main__.trainvae`` (``__main__.py:1065-1107``) runs unchanged on top of them.

Division of labour
  * host (this file): feature normalisation with numpy exactly as the reference does it
    (encode.py:98-126; the DataLoader object is the reference's own boundary type), the epoch /
    batch-size schedule (359-440, 543-610), shuffling, logging, ``model.pt`` (486-541).
  * device (``csrc/vae.hip`` through the C ABI): everything inside the batch loop -- forward, loss,
    backward, D-Adapt-Adam -- and the eval-mode encode pass.  The normalised feature matrix is
    uploaded once and stays resident in HBM; there is one host synchronisation per epoch.
"""
from __future__ import annotations

import ctypes
import logging
import weakref
from collections import OrderedDict
from math import log as _log
from pathlib import Path
from typing import IO, Optional, Union

import numpy as _np
import torch as _torch
from torch.utils.data import DataLoader as _DataLoader
from torch.utils.data.dataset import TensorDataset as _TensorDataset

from . import _lib

try:   # the reference logs through loguru (encode.py:11): after dropin.install() the epoch lines land in vamb's log
    from loguru import logger
except ImportError:   # pragma: no cover - loguru is a dependency of vamb, not of this package
    logger = logging.getLogger("vamb_amd.encode")
NTNF = 103

# Arithmetic of the dense contractions of VAEs created from now on: "fp32" (fp32 MFMA, BASELINE config C1; the
# default) or "bf16" (operands rounded to bf16, fp32 accumulation; BASELINE configs C2-C4).  The reference
# signature of VAE() has no such argument, so it is a module setting (environment override: VAMBHIP_PRECISION).
_COMPUTE_DTYPE = "fp32"


def set_compute_dtype(dtype: str) -> None:
    global _COMPUTE_DTYPE
    if dtype not in ("fp32", "bf16"):
        raise ValueError(f"compute dtype must be 'fp32' or 'bf16', not {dtype!r}")
    _COMPUTE_DTYPE = dtype


def get_compute_dtype() -> str:
    import os

    return os.environ.get("VAMBHIP_PRECISION", _COMPUTE_DTYPE)


# Where make_dataloader normalises the features: "host" (numpy, exactly the reference's statements), "device"
# (the matrix passes as HIP kernels over ONE upload of the raw matrices, csrc/prep.hip; bit-identical tensors) or
# "auto" (device when a GPU is visible and the inputs are C-contiguous).  Environment override: VAMBHIP_PREP.
_PREP_MODE = "auto"


def set_prep_mode(mode: str) -> None:
    global _PREP_MODE
    if mode not in ("auto", "host", "device"):
        raise ValueError(f"prep mode must be 'auto', 'host' or 'device', not {mode!r}")
    _PREP_MODE = mode


def get_prep_mode() -> str:
    import os

    return os.environ.get("VAMBHIP_PREP", _PREP_MODE)


def _pairwise_program(n: int) -> _np.ndarray:
    """numpy's pairwise summation of n contiguous float32 values (numpy/_core/src/umath/loops_utils.h.src,
    ``@TYPE@_pairwise_sum``) as a postfix program of int32 triples: (0, start, len) pushes the sum of a block of
    at most 128 elements (eight interleaved accumulators, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
    len % 8 tail one by one; a plain loop below 8 elements), (1, 0, 0) adds the two topmost values.  Blocks larger
    than 128 are split at n // 2 rounded down to a multiple of 8.  ``a.sum(axis=1)`` of a C-contiguous float32
    matrix is 0 + this sum for every row (checked against numpy in tests/test_prep_host.py)."""
    ops = []

    def rec(start: int, length: int) -> None:
        if length <= 128:
            ops.append((0, start, length))
            return
        half = length // 2
        half -= half % 8
        rec(start, half)
        rec(start + half, length - half)
        ops.append((1, 0, 0))

    rec(0, int(n))
    return _np.asarray(ops, dtype=_np.int32).reshape(-1, 3)


def _zscore_inplace(array: _np.ndarray, axis: Optional[int] = None) -> None:
    """In-place z-score with the reference's conventions (vambtools.py:250-288): population std,
    zero std replaced by 1."""
    mean = array.mean(axis=axis)
    std = array.std(axis=axis)
    if axis is None:
        if std == 0:
            std = 1
    else:
        std[std == 0.0] = 1
        shape = tuple(dim if ax != axis else 1 for ax, dim in enumerate(array.shape))
        mean.shape, std.shape = shape, shape
    array -= mean
    array /= std


def set_batchsize(data_loader: _DataLoader, batch_size: int, n_obs: int, encode=False) -> _DataLoader:
    """Copy of the data loader with another batch size (encode.py:33-50).  ``encode=True`` gives the
    ordered, keep-everything loader used before encoding."""
    return _DataLoader(
        dataset=data_loader.dataset,
        batch_size=batch_size,
        shuffle=not encode,
        drop_last=not encode and (n_obs > batch_size),
        num_workers=0,
        pin_memory=data_loader.pin_memory,
        collate_fn=data_loader.collate_fn,
    )


def make_dataloader(abundance: _np.ndarray, tnf: _np.ndarray, lengths: _np.ndarray, batchsize: int = 256,
                    destroy: bool = False, cuda: bool = False, _prep: Optional[str] = None) -> _DataLoader:
    """Normalise abundance / TNF / lengths and wrap them as the reference's DataLoader
    (encode.py:53-146): tensors are (depths [N,S], tnf [N,103], total_abundance [N,1], weights [N,1]).
    ``_prep`` (private): "host" / "device" for THIS call, whatever the module setting and VAMBHIP_PREP say."""
    if not isinstance(abundance, _np.ndarray) or not isinstance(tnf, _np.ndarray):
        raise ValueError("TNF and abundance must be Numpy arrays")
    if batchsize < 1:
        raise ValueError(f"Batch size must be minimum 1, not {batchsize}")
    if len(abundance) != len(tnf) or len(tnf) != len(lengths):
        raise ValueError("Lengths of abundance, TNF and lengths arrays must be the same")
    if not (abundance.dtype == tnf.dtype == _np.float32):
        raise ValueError("TNF and abundance must be Numpy arrays of dtype float32")
    mode = get_prep_mode() if _prep is None else _prep
    if mode not in ("auto", "host", "device"):
        raise ValueError(f"prep mode must be 'auto', 'host' or 'device', not {mode!r}")
    if mode != "host" and _device_prep_possible(abundance, tnf, required=(mode == "device")):
        return _make_dataloader_device(abundance, tnf, lengths, batchsize, destroy)
    if not destroy:
        abundance = abundance.copy()
        tnf = tnf.copy()

    # equal sequencing depth per sample, then per-contig total and relative abundance (98-113)
    sample_depths_sum = abundance.sum(axis=0)
    if _np.any(sample_depths_sum == 0):
        raise ValueError("One or more samples have zero depth in all sequences, so cannot be depth normalized")
    abundance *= 1_000_000 / sample_depths_sum
    total_abundance = abundance.sum(axis=1)
    n_samples = abundance.shape[1]
    zero_total = total_abundance == 0
    abundance[zero_total] = 1 / n_samples
    divisor = total_abundance.copy()
    divisor[zero_total] = 1.0
    abundance /= divisor.reshape((-1, 1))

    # log total abundance and TNF are z-scored (116-119)
    total_abundance = _np.log(total_abundance.clip(min=0.001))
    _zscore_inplace(total_abundance)
    _zscore_inplace(tnf, axis=0)
    total_abundance.shape = (len(total_abundance), 1)

    # contig weights from lengths (122-126)
    weights = _weights_from_lengths(lengths)

    dataset = _TensorDataset(_torch.from_numpy(abundance), _torch.from_numpy(tnf),
                             _torch.from_numpy(total_abundance), _torch.from_numpy(weights))
    # The loader is only a container here (the batch loop runs on the GPU over the resident matrix),
    # so no worker processes are spawned.
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=(len(abundance) > batchsize),
                       shuffle=True, num_workers=0, pin_memory=False)


def _weights_from_lengths(lengths: _np.ndarray) -> _np.ndarray:
    """Contig weights (encode.py:122-126)."""
    lengths = lengths.astype(_np.float32)
    weights = _np.log(lengths).astype(_np.float32) - 5.0
    weights[weights < 2.0] = 2.0
    weights *= len(weights) / weights.sum()
    weights.shape = (len(weights), 1)
    return weights


def _device_prep_possible(abundance: _np.ndarray, tnf: _np.ndarray, required: bool) -> bool:
    why = None
    if not (abundance.flags.c_contiguous and tnf.flags.c_contiguous):
        why = "abundance / tnf are not C-contiguous (numpy's summation order depends on the layout)"
    elif abundance.ndim != 2 or tnf.ndim != 2 or tnf.shape[1] != NTNF or abundance.shape[1] < 1 or len(abundance) < 1:
        why = "unexpected matrix shapes"
    else:
        try:
            if _lib.device_count() < 1:
                why = "no HIP device visible"
        except Exception as e:   # library not built / no driver
            why = str(e)
    if why is None:
        return True
    if required:
        raise _lib.VambHipError(f"device feature preparation requested but not possible: {why}")
    return False


class _LazyHostTensor:
    """Stand-in for one tensor of a device-prepared dataset.  ``.shape`` / ``len()`` / ``.size()`` / ``.dtype`` -- all the
    reference's drivers ever read (``data_loader.dataset.tensors[0].shape``, vamb/__main__.py:1073-1074,1119,1135) --
    cost nothing; anything that needs the VALUES (indexing, ``.numpy()``, any torch function) downloads the dataset from
    HBM once and behaves like the real ``torch.Tensor`` from then on."""

    __slots__ = ("_owner", "_index", "shape")

    def __init__(self, owner, index: int, shape):
        self._owner = owner
        self._index = index
        self.shape = _torch.Size(shape)

    dtype = _torch.float32
    device = _torch.device("cpu")

    def _real(self):
        return self._owner._materialise()[self._index]

    def __len__(self) -> int:
        return self.shape[0]

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self) -> int:
        return len(self.shape)

    ndim = property(dim)

    def __getitem__(self, index):
        return self._real()[index]

    def __iter__(self):
        return iter(self._real())

    def __getattr__(self, name):   # only reached for names not defined above
        return getattr(self._real(), name)

    def __repr__(self) -> str:
        return f"<device-resident float32 tensor of shape {tuple(self.shape)} (host copy made on first element access)>"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def real(x):
            if isinstance(x, _LazyHostTensor):
                return x._real()
            if isinstance(x, (list, tuple)):
                return type(x)(real(y) for y in x)
            return x

        return func(*real(args), **{k: real(v) for k, v in (kwargs or {}).items()})


class _PreparedDataset(_torch.utils.data.Dataset):
    """The TensorDataset of make_dataloader when the features were normalised on the device: the four tensors live
    in HBM (``vh_dataset``).  ``.tensors`` is a tuple of shape-only proxies (``_LazyHostTensor``): reading a shape never
    moves data; host copies are made on the first access to an element (the reference's callers only hand the loader to
    VAE.trainmodel / VAE.encode, which never need them)."""

    def __init__(self, lib, handle, n: int, nsamples: int):
        self._lib = lib
        self.handle = handle
        self.n = int(n)
        self.nsamples = int(nsamples)
        self._tensors = None
        shapes = ((self.n, self.nsamples), (self.n, NTNF), (self.n, 1), (self.n, 1))
        self._proxies = tuple(_LazyHostTensor(self, i, sh) for i, sh in enumerate(shapes))

    @property
    def _vambhip_prepared(self):
        return self

    def _materialise(self, into=None):
        """Host copies of the four tensors (one download).  ``into`` = (abundance, tnf): the caller's own C-contiguous arrays
        receive the normalised blocks (make_dataloader(destroy=True)) and back the host tensors, as torch.from_numpy does in
        the reference."""
        if self._tensors is None:
            d = into[0] if into is not None else _np.empty((self.n, self.nsamples), _np.float32)
            t = into[1] if into is not None else _np.empty((self.n, NTNF), _np.float32)
            a = _np.empty((self.n, 1), _np.float32)
            w = _np.empty((self.n, 1), _np.float32)
            _lib.check(self._lib.vh_dataset_download(self.handle, _lib.ptr(d), _lib.ptr(t), _lib.ptr(a), _lib.ptr(w)))
            self._tensors = tuple(_torch.from_numpy(x) for x in (d, t, a, w))
        return self._tensors

    @property
    def tensors(self):
        return self._tensors if self._tensors is not None else self._proxies

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, index):
        return tuple(t[index] for t in self._materialise())

    def __del__(self):
        try:
            if self.handle is not None:
                self._lib.vh_dataset_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _make_dataloader_device(abundance: _np.ndarray, tnf: _np.ndarray, lengths: _np.ndarray, batchsize: int,
                            destroy: bool = False) -> _DataLoader:
    """make_dataloader (encode.py:98-126) with the O(N x columns) passes on the GPU.  The statements below are the
    reference's, in its order; wherever it reduces or rescales a whole matrix the vector comes from / goes to a kernel
    of csrc/prep.hip that reproduces numpy's float32 summation order, so the tensors are bit-identical to the host
    path (tests/test_prep_gpu.py).  The raw arrays are uploaded once.  ``destroy=False``: they are never modified.
    ``destroy=True`` (encode.py:86-89: "mutate abundance and tnf array in-place"): the normalised blocks are written back
    into the caller's arrays and the dataset's host tensors ARE those arrays, as in the reference
    (test/test_encode.py:47-95: test_destroy, test_normalized, test_single_sample)."""
    lib = _lib.load()
    n, n_samples = abundance.shape
    h = ctypes.c_void_p()
    _lib.check(lib.vh_prep_create(n, n_samples, ctypes.byref(h)))
    try:
        _lib.check(lib.vh_prep_upload(h, _lib.ptr(abundance), _lib.ptr(tnf)))
        # sample_depths_sum = abundance.sum(axis=0)
        if n_samples == 1:
            # a single column is a contiguous 1-d reduction to numpy (PAIRWISE summation over the rows, not the row-order
            # chain it runs for wider matrices): 4 n bytes, taken on the host from the caller's array
            sample_depths_sum = abundance.sum(axis=0)
        else:
            sample_depths_sum = _np.empty(n_samples, _np.float32)
            _lib.check(lib.vh_prep_column_sums(h, 0, None, _lib.ptr(sample_depths_sum)))
        if _np.any(sample_depths_sum == 0):
            raise ValueError("One or more samples have zero depth in all sequences, so cannot be depth normalized")
        scale = _np.ascontiguousarray(1_000_000 / sample_depths_sum, dtype=_np.float32)
        # abundance *= scale; total_abundance = abundance.sum(axis=1); zero rows -> 1 / n_samples; abundance /= total
        program = _np.ascontiguousarray(_pairwise_program(n_samples))
        total_abundance = _np.empty(n, _np.float32)
        _lib.check(lib.vh_prep_normalise_rows(h, _lib.ptr(scale), _lib.ptr(program), len(program),
                                              ctypes.c_float(_np.float32(1 / n_samples)), _lib.ptr(total_abundance)))
        total_abundance = _np.log(total_abundance.clip(min=0.001))
        _zscore_inplace(total_abundance)
        # zscore(tnf, axis=0): mean and std exactly as numpy's _mean / _var build them from the column sums
        count = _np.intp(n)
        mean = _np.empty(NTNF, _np.float32)
        _lib.check(lib.vh_prep_column_sums(h, 1, None, _lib.ptr(mean)))
        mean = _np.true_divide(mean, count, out=mean, casting="unsafe")
        var = _np.empty(NTNF, _np.float32)
        _lib.check(lib.vh_prep_column_sums(h, 1, _lib.ptr(mean), _lib.ptr(var)))
        var = _np.true_divide(var, _np.maximum(count - 0, 0), out=var, casting="unsafe")
        std = _np.sqrt(var, out=var)
        std[std == 0.0] = 1
        _lib.check(lib.vh_prep_zscore_tnf(h, _lib.ptr(mean), _lib.ptr(std)))
        weights = _weights_from_lengths(lengths)
        d = ctypes.c_void_p()
        _lib.check(lib.vh_prep_finish(h, _lib.ptr(_np.ascontiguousarray(total_abundance)),
                                      _lib.ptr(_np.ascontiguousarray(weights.reshape(-1))), ctypes.byref(d)))
    finally:
        lib.vh_prep_destroy(h)
    dataset = _PreparedDataset(lib, d, n, n_samples)
    if destroy:
        dataset._materialise(into=(abundance, tnf))
    return _DataLoader(dataset=dataset, batch_size=batchsize, drop_last=(n > batchsize), shuffle=True, num_workers=0,
                       pin_memory=False)


class _Config(ctypes.Structure):
    _fields_ = [("nsamples", ctypes.c_int32), ("nlatent", ctypes.c_int32), ("nlayers", ctypes.c_int32),
                ("nhiddens", ctypes.c_int32 * 8), ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
                ("dropout", ctypes.c_float), ("seed", ctypes.c_uint64)]


def _as_f32(x) -> _np.ndarray:
    if isinstance(x, _torch.Tensor):
        x = x.detach().cpu().numpy()
    return _np.ascontiguousarray(x, dtype=_np.float32)


class _DeviceDataset:
    """Feature matrix + weights resident in HBM (vh_dataset), keyed by the host tensors it was made from."""

    def __init__(self, lib, key, d, t, a, w, n, nsamples):
        self._lib = lib
        self.key = key
        h = ctypes.c_void_p()
        _lib.check(lib.vh_dataset_create(_lib.ptr(d), _lib.ptr(t), _lib.ptr(a), _lib.ptr(w), n, nsamples, ctypes.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if self.handle is not None:
                self._lib.vh_dataset_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class VAE:
    """Variational autoencoder with the reference's interface (encode.py:149-257); the network lives
    on the GPU behind ``libvambhip``.

    Instantiate with:
        nsamples: Number of samples in abundance matrix
        nhiddens: list of n_neurons in the hidden layers [None=Auto]
        nlatent: Number of neurons in the latent layer [32]
        alpha: Approximate starting TNF/(CE+TNF) ratio in loss. [None = Auto]
        beta: Multiply KLD by the inverse of this value [200]
        dropout: Probability of dropout on forward pass [0.2]
        cuda: accepted for compatibility; the model always runs on the GPU
        seed: seeds parameter initialisation, shuffling, dropout and noise
    """

    def __init__(self, nsamples: int, nhiddens: Optional[list[int]] = None, nlatent: int = 32,
                 alpha: Optional[float] = None, beta: float = 200.0, dropout: Optional[float] = 0.2,
                 cuda: bool = False, seed: int = 0):
        if nlatent < 1:
            raise ValueError(f"Minimum 1 latent neuron, not {nlatent}")
        if nsamples < 1:
            raise ValueError(f"nsamples must be > 0, not {nsamples}")
        if alpha is None:
            alpha = 0.15 if nsamples > 1 else 0.50
        if nhiddens is None:
            nhiddens = [512, 512] if nsamples > 1 else [256, 256]
        if dropout is None:
            dropout = 0.2 if nsamples > 1 else 0.0
        if any(i < 1 for i in nhiddens):
            raise ValueError(f"Minimum 1 neuron per layer, not {min(nhiddens)}")
        if beta <= 0:
            raise ValueError(f"beta must be > 0, not {beta}")
        if not (0 < alpha < 1):
            raise ValueError(f"alpha must be 0 < alpha < 1, not {alpha}")
        if not (0 <= dropout < 1):
            raise ValueError(f"dropout must be 0 <= dropout < 1, not {dropout}")
        if len(nhiddens) > 8:
            raise ValueError("at most 8 hidden layers are supported")

        _torch.manual_seed(seed)
        self.usecuda = True
        self.nsamples = nsamples
        self.ntnf = NTNF
        self.alpha = alpha
        self.beta = beta
        self.nhiddens = list(nhiddens)
        self.nlatent = nlatent
        self.dropout = dropout
        self.training = True  # a fresh torch module is in train() mode

        self._lib = _lib.load()
        _lib.require_gpu()
        cfg = _Config()
        # (a subclass whose super().__init__ argument is not the number of samples sets _native_nsamples first)
        cfg.nsamples, cfg.nlatent, cfg.nlayers = getattr(self, "_native_nsamples", nsamples), nlatent, len(nhiddens)
        for i, n in enumerate(nhiddens):
            cfg.nhiddens[i] = n
        cfg.alpha, cfg.beta, cfg.dropout, cfg.seed = alpha, beta, dropout, seed & 0xFFFFFFFFFFFFFFFF
        _lib.sync_env_options()   # VAMBHIP_* variables -> library options (the .so reads no environment)
        self._h = self._create_handle(cfg)
        self._dataset_key = None
        self._dataset_ref = None
        self._n_rows = 0
        self._comm = None
        self.compute_dtype = get_compute_dtype()
        if self.compute_dtype not in ("fp32", "bf16"):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', not {self.compute_dtype!r}")
        try:
            _lib.check(self._lib.vh_vae_set_precision(self._h, int(self.compute_dtype == "bf16")))
        except ValueError as e:
            # too wide for the bf16 step (a label block of several thousand classes): the fp32 step has no such limit
            import warnings

            warnings.warn(f"vamb_amd: {e}; this model runs the fp32 step", RuntimeWarning, stacklevel=2)
            self.compute_dtype = "fp32"
            _lib.check(self._lib.vh_vae_set_precision(self._h, 0))

    def _create_handle(self, cfg) -> ctypes.c_void_p:
        h = ctypes.c_void_p()
        _lib.check(self._lib.vh_vae_create(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def _row_width(self) -> int:
        """Input (= reconstruction) columns: depths | TNF | total abundance."""
        return self.nsamples + NTNF + 1

    def attach_communicator(self, comm, syncbn: bool = True) -> None:
        """Data-parallel training over ``comm`` (``vamb_amd.parallel.Communicator``): this process holds
        one row shard of the dataset; every step all-reduces the gradient over RCCL inside the library.
        ``syncbn`` (default): BatchNorm batch statistics span the all-rank batch, i.e. the step computes what the
        single-process reference computes on the whole batch; False = per-rank statistics (fewer collectives)."""
        self._comm = comm
        _lib.check(self._lib.vh_vae_attach_comm(self._h, comm.handle if comm is not None else None))
        _lib.check(self._lib.vh_vae_set_syncbn(self._h, int(bool(syncbn))))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.vh_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- torch.nn.Module look-alikes ------------------------------------------------------------
    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def cuda(self):
        return self

    def _state_names(self):
        names = []
        nl = len(self.nhiddens)
        for kind in ("encoder", "decoder"):
            for i in range(nl):
                names += [f"{kind}layers.{i}.weight", f"{kind}layers.{i}.bias"]
            for i in range(nl):
                names += [f"{kind}norms.{i}.{s}" for s in
                          ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
        names += ["mu.weight", "mu.bias", "outputlayer.weight", "outputlayer.bias"]
        return names

    def _shape_of(self, name: str):
        d = self._row_width()
        enc_in = [d] + self.nhiddens[:-1]
        dec_w = self.nhiddens[::-1]
        dec_in = [self.nlatent] + dec_w[:-1]
        part, rest = name.split(".", 1)
        if part in ("mu", "outputlayer"):
            nout, nin = (self.nlatent, self.nhiddens[-1]) if part == "mu" else (d, self.nhiddens[0])
            return (nout, nin) if rest == "weight" else (nout,)
        idx, field = rest.split(".", 1)
        idx = int(idx)
        if part == "encoderlayers":
            return (self.nhiddens[idx], enc_in[idx]) if field == "weight" else (self.nhiddens[idx],)
        if part == "decoderlayers":
            return (dec_w[idx], dec_in[idx]) if field == "weight" else (dec_w[idx],)
        width = self.nhiddens[idx] if part == "encodernorms" else dec_w[idx]
        return () if field == "num_batches_tracked" else (width,)

    def state_dict(self) -> "OrderedDict[str, _torch.Tensor]":
        out = OrderedDict()
        for name in self._state_names():
            shape = self._shape_of(name)
            n = int(_np.prod(shape)) if shape else 1
            buf = _np.empty(n, _np.float32)
            _lib.check(self._lib.vh_vae_get_param(self._h, name.encode(), _lib.ptr(buf), n))
            if name.endswith("num_batches_tracked"):
                out[name] = _torch.tensor(int(buf[0]), dtype=_torch.int64)
            else:
                out[name] = _torch.from_numpy(buf.reshape(shape).copy())
        return out

    def load_state_dict(self, state) -> None:
        expected = set(self._state_names())
        got = set(state.keys())
        if expected != got:
            raise RuntimeError(f"state_dict mismatch: missing {sorted(expected - got)}, "
                               f"unexpected {sorted(got - expected)}")
        for name in self._state_names():
            value = state[name]
            arr = _as_f32(value).reshape(-1)
            shape = self._shape_of(name)
            n = int(_np.prod(shape)) if shape else 1
            if arr.size != n:
                raise RuntimeError(f"size mismatch for {name}: expected {shape}, got {tuple(_np.shape(value))}")
            _lib.check(self._lib.vh_vae_set_param(self._h, name.encode(), _lib.ptr(arr), n))

    def parameters_gradient(self, name: str) -> _np.ndarray:
        """Gradient of the last training step for one parameter (what ``p.grad`` holds after
        ``loss.backward()`` in the reference, encode.py:418)."""
        shape = self._shape_of(name)
        n = int(_np.prod(shape))
        buf = _np.empty(n, _np.float32)
        _lib.check(self._lib.vh_vae_get_grad(self._h, name.encode(), _lib.ptr(buf), n))
        return buf.reshape(shape)

    def hidden_activations(self, layer: int, batch: int) -> _np.ndarray:
        """Post-dropout activations of hidden layer ``layer`` (encoder layers first) for the last training batch."""
        width = self.nhiddens[layer] if layer < len(self.nhiddens) else list(reversed(self.nhiddens))[layer - len(self.nhiddens)]
        buf = _np.empty((batch, width), _np.float32)
        _lib.check(self._lib.vh_vae_get_hidden(self._h, int(layer), _lib.ptr(buf), buf.size))
        return buf

    def optimizer_state(self):
        d, nw, k = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._lib.vh_vae_opt_state(self._h, ctypes.byref(d), ctypes.byref(nw), ctypes.byref(k)))
        return dict(d=d.value, numerator_weighted=nw.value, k=k.value)

    # ---- forward / loss (encode.py:306-357) -------------------------------------------------------
    def forward(self, depths, tnf, abundance, _eps=None, _masks=None):
        d, t, a = _as_f32(depths), _as_f32(tnf), _as_f32(abundance)
        if d.ndim != 2 or d.shape[1] != self.nsamples or t.shape != (len(d), NTNF) or a.shape != (len(d), 1):
            raise ValueError("expected depths [B, nsamples], tnf [B, 103], abundance [B, 1]")
        b = len(d)
        do = _np.empty((b, self.nsamples), _np.float32)
        to = _np.empty((b, NTNF), _np.float32)
        ao = _np.empty((b, 1), _np.float32)
        mu = _np.empty((b, self.nlatent), _np.float32)
        eps = None if _eps is None else _as_f32(_eps)
        masks = None if _masks is None else _np.ascontiguousarray(
            _np.concatenate([_np.asarray(m, dtype=_np.uint8).reshape(-1) for m in _masks]))
        _lib.check(self._lib.vh_vae_forward(self._h, _lib.ptr(d), _lib.ptr(t), _lib.ptr(a), b, int(self.training),
                                            _lib.ptr(eps), _lib.ptr(masks), _lib.ptr(do), _lib.ptr(to),
                                            _lib.ptr(ao), _lib.ptr(mu)))
        return (_torch.from_numpy(do), _torch.from_numpy(to), _torch.from_numpy(ao), _torch.from_numpy(mu))

    __call__ = forward

    def calc_loss(self, depths_in, depths_out, tnf_in, tnf_out, abundance_in, abundance_out, mu, weights):
        """The reference's loss on host tensors (encode.py:316-357), for callers that evaluate
        ``vae(...)`` outputs themselves (test_encode.py:159-167).  Training computes the same
        quantities on the device.  Note ``weights`` is [B,1] while the row terms are [B]: like the
        reference this broadcasts to [B,B] before ``.mean()``."""
        t = lambda x: x if isinstance(x, _torch.Tensor) else _torch.as_tensor(x)  # noqa: E731
        depths_in, depths_out, tnf_in, tnf_out = t(depths_in), t(depths_out), t(tnf_in), t(tnf_out)
        abundance_in, abundance_out, mu, weights = t(abundance_in), t(abundance_out), t(mu), t(weights)
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
        weighed_ab, weighed_ce = ab_sse * ab_sse_weight, ce * ce_weight
        weighed_sse, weighed_kld = sse * sse_weight, kld * kld_weight
        loss = ((weighed_ce + weighed_ab + weighed_sse) + weighed_kld) * weights
        return (loss.mean(), weighed_ab.mean(), weighed_ce.mean(), weighed_sse.mean(), weighed_kld.mean())

    # ---- dataset residency --------------------------------------------------------------------------
    def _ensure_dataset(self, data_loader) -> int:
        """Make the loader's four tensors resident in HBM (once per loader: the device copy is cached on the
        dataset object and shared by every VAE trained / encoded on it)."""
        prepared = getattr(data_loader.dataset, "_vambhip_prepared", None)
        if prepared is not None:     # normalised on the device by make_dataloader: already resident
            if prepared.nsamples != self.nsamples:
                raise ValueError("dataset tensors do not match this VAE (nsamples / 103 TNF / 1 / 1 columns)")
            key = ("prepared", id(prepared))
            if self._dataset_ref is None or self._dataset_ref() is not prepared or self._dataset_key != key:
                _lib.check(self._lib.vh_vae_use_dataset(self._h, prepared.handle))
                self._device_dataset = prepared
                self._dataset_key = key
                self._dataset_ref = weakref.ref(prepared)
                self._n_rows = prepared.n
            return prepared.n
        tensors = data_loader.dataset.tensors
        if len(tensors) != 4:
            raise ValueError("expected a DataLoader made by make_dataloader (4 tensors)")
        holder = data_loader.dataset
        # identity of the host data: the dataset OBJECT (a weak reference: a new dataset allocated at a freed address
        # is a different object), the tensors' storage and their in-place modification counters
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in tensors)
        n = len(tensors[0])
        same_object = self._dataset_ref is not None and self._dataset_ref() is holder
        if not same_object or key != self._dataset_key:
            d, t, a, w = (_as_f32(x) for x in tensors)
            if d.shape != (n, self.nsamples) or t.shape != (n, NTNF) or a.shape != (n, 1) or w.shape != (n, 1):
                raise ValueError("dataset tensors do not match this VAE (nsamples / 103 TNF / 1 / 1 columns)")
            cached = getattr(holder, "_vambhip_device", None)
            if cached is None or cached.key != key:
                cached = _DeviceDataset(self._lib, key, d, t, a, w, n, self.nsamples)
                try:
                    holder._vambhip_device = cached
                except AttributeError:
                    pass
            _lib.check(self._lib.vh_vae_use_dataset(self._h, cached.handle))
            self._device_dataset = cached      # keeps the shared device copy alive while this VAE uses it
            self._dataset_key = key
            self._dataset_ref = weakref.ref(holder)
            self._n_rows = n
        return n

    def invalidate_dataset(self) -> None:
        """Forget the device copy of the dataset: the next train / encode call uploads the host tensors again."""
        self._dataset_key = None
        self._dataset_ref = None

    def train_batch(self, rows, eps=None, masks=None):
        """One optimisation step on explicit dataset rows with optionally injected randomness
        (parity tests).  Returns the five means of calc_loss: (loss, ab_sse, ce, sse, kld)."""
        rows = _np.ascontiguousarray(rows, dtype=_np.int64)
        e = None if eps is None else _as_f32(eps)
        m = None if masks is None else _np.ascontiguousarray(
            _np.concatenate([_np.asarray(x, dtype=_np.uint8).reshape(-1) for x in masks]))
        out = (ctypes.c_double * 5)()
        _lib.check(self._lib.vh_vae_train_step(self._h, _lib.ptr(rows), len(rows), _lib.ptr(e), _lib.ptr(m), out))
        return tuple(out)

    # ---- training (encode.py:359-440, 543-610) ------------------------------------------------------
    def trainepoch(self, data_loader, epoch: int, optimizer, batchsteps: list[int]):
        """One epoch (encode.py:359-440).  `optimizer` is accepted for signature compatibility and NOT used: the
        D-Adapt-Adam state lives in the native handle (reset by trainmodel, readable through optimizer_state())."""
        n_seq = self._ensure_dataset(data_loader)
        if n_seq < 2:
            raise ValueError(
                f"Cannot train on a dataset with fewer than 2 sequences, but got {n_seq} sequences. "
                "If you are trying to fit a DL model to this few sequences, "
                "something probably went wrong in your pipeline.")
        self.train()
        if epoch in batchsteps:
            data_loader = set_batchsize(data_loader, data_loader.batch_size * 2, n_seq)
        bs = data_loader.batch_size
        means = (ctypes.c_double * 5)()
        # RandomSampler semantics: a fresh shuffle per epoch, the ragged tail dropped iff n > batch.  The
        # shuffle itself happens on the device (perm = NULL): no host randperm, no upload.
        if self._comm is not None:
            # data parallel: `bs` is the ALL-RANK batch; this rank contributes bs / world local rows and
            # every rank must run the same number of collective steps
            world = self._comm.world
            if bs % world != 0:
                raise ValueError(f"global batch {bs} is not divisible by the number of GPUs {world}")
            batch = bs // world
            if n_seq < batch:
                raise ValueError(f"shard of {n_seq} rows is smaller than the per-GPU batch {batch}")
            key = (n_seq, batch)
            if getattr(self, "_dp_plan_key", None) != key:
                nb = self._comm.all_reduce_min(_np.array([n_seq // batch], dtype=_np.int64))
                self._dp_plan_key, self._dp_batches = key, int(nb[0])
            n_batches = self._dp_batches
            _lib.check(self._lib.vh_vae_train_epoch_dp(self._h, None, n_batches, batch, bs, None, means))
        else:
            if n_seq > bs:
                n_batches = n_seq // bs
                batch = bs
            else:
                n_batches, batch = 1, n_seq
            _lib.check(self._lib.vh_vae_train_epoch(self._h, None, n_batches, batch, means))
        loss, ab, ce, sse, kld = tuple(means)
        logger.info(
            "\t\tEpoch: {:>3}  Loss: {:.5e}  CE: {:.5e}  AB: {:.5e}  SSE: {:.5e}  KLD: {:.5e}  Batchsize: {:>4}".format(
                epoch + 1, loss, ce, ab, sse, kld, bs))
        self.last_epoch_losses = dict(loss=loss, ce=ce, ab=ab, sse=sse, kld=kld, batchsize=bs)
        self.eval()
        return data_loader

    def _train_segment(self, data_loader, first_epoch: int, count: int, batchsteps) -> "_DataLoader":
        """`count` epochs starting at `first_epoch`, none of which except the first changes the batch size."""
        n_seq = self._ensure_dataset(data_loader)
        if n_seq < 2:
            raise ValueError(
                f"Cannot train on a dataset with fewer than 2 sequences, but got {n_seq} sequences. "
                "If you are trying to fit a DL model to this few sequences, "
                "something probably went wrong in your pipeline.")
        self.train()
        if first_epoch in batchsteps:
            data_loader = set_batchsize(data_loader, data_loader.batch_size * 2, n_seq)
        bs = data_loader.batch_size
        means = (ctypes.c_double * (5 * count))()
        if self._comm is not None:
            world = self._comm.world
            if bs % world != 0:
                raise ValueError(f"global batch {bs} is not divisible by the number of GPUs {world}")
            batch = bs // world
            if n_seq < batch:
                raise ValueError(f"shard of {n_seq} rows is smaller than the per-GPU batch {batch}")
            key = (n_seq, batch)
            if getattr(self, "_dp_plan_key", None) != key:
                nb = self._comm.all_reduce_min(_np.array([n_seq // batch], dtype=_np.int64))
                self._dp_plan_key, self._dp_batches = key, int(nb[0])
            n_batches, global_batch = self._dp_batches, bs
        else:
            if n_seq > bs:
                n_batches, batch = n_seq // bs, bs
            else:
                n_batches, batch = 1, n_seq
            global_batch = 0
        _lib.check(self._lib.vh_vae_train_epochs(self._h, count, n_batches, batch, global_batch, means))
        for e in range(count):
            loss, ab, ce, sse, kld = tuple(means[5 * e: 5 * e + 5])
            logger.info(
                "\t\tEpoch: {:>3}  Loss: {:.5e}  CE: {:.5e}  AB: {:.5e}  SSE: {:.5e}  KLD: {:.5e}  Batchsize: {:>4}".format(
                    first_epoch + e + 1, loss, ce, ab, sse, kld, bs))
        self.last_epoch_losses = dict(loss=loss, ce=ce, ab=ab, sse=sse, kld=kld, batchsize=bs)
        self.eval()
        return data_loader

    def trainmodel(self, dataloader, nepochs: int = 500, batchsteps: Optional[list[int]] = [25, 75, 150, 300],
                   modelfile: Union[None, str, Path, IO[bytes]] = None):
        """Train the autoencoder (encode.py:543-610).  Output: None"""
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
        prepared = getattr(dataloader.dataset, "_vambhip_prepared", None)
        if prepared is not None:
            ncontigs, nsamples = prepared.n, prepared.nsamples
        else:
            ncontigs, nsamples = dataloader.dataset.tensors[0].shape
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
        logger.info(f"\t    N sequences: {ncontigs}")
        logger.info(f"\t    N samples: {nsamples}")
        # a fresh optimiser per call, as the reference's `DAdaptAdam(self.parameters(), decouple=True)` (encode.py:578) -- also
        # for a network the joint TaxVamb trainer has driven with Adam before
        _lib.check(self._lib.vh_vae_set_optimizer(self._h, 0, 0.0))   # VH_OPT_DADAPT_ADAM
        _lib.check(self._lib.vh_vae_reset_optimizer(self._h))
        # the epochs between two batch-size changes are enqueued by ONE library call (a single host
        # synchronisation per segment instead of one per epoch); the log lines are those of trainepoch
        epoch = 0
        while epoch < nepochs:
            nxt = min([b for b in batchsteps_set if b > epoch] + [nepochs])
            dataloader = self._train_segment(dataloader, epoch, nxt - epoch, batchsteps_set)
            epoch = nxt
        if modelfile is not None:
            try:
                self.save(modelfile)
            except Exception:
                pass
        return None

    # ---- encode (encode.py:442-484) -----------------------------------------------------------------
    def encode(self, data_loader) -> _np.ndarray:
        self.eval()
        n = self._ensure_dataset(data_loader)
        latent = _np.empty((n, self.nlatent), dtype=_np.float32)
        _lib.check(self._lib.vh_vae_encode(self._h, _lib.ptr(latent)))
        return latent

    # ---- model.pt (encode.py:486-541) ---------------------------------------------------------------
    def save(self, filehandle):
        state = {"nsamples": self.nsamples, "alpha": self.alpha, "beta": self.beta, "dropout": self.dropout,
                 "nhiddens": self.nhiddens, "nlatent": self.nlatent, "state": self.state_dict()}
        _torch.save(state, filehandle)

    @classmethod
    def load(cls, path: Union[IO[bytes], str], cuda: bool = False, evaluate: bool = True):
        dictionary = _torch.load(path, map_location=lambda storage, loc: storage, weights_only=True)
        vae = cls(dictionary["nsamples"], dictionary["nhiddens"], dictionary["nlatent"], dictionary["alpha"],
                  dictionary["beta"], dictionary["dropout"], cuda)
        vae.load_state_dict(dictionary["state"])
        if evaluate:
            vae.eval()
        return vae
