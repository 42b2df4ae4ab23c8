"""Tetranucleotide frequencies on MI355X -- the compute of ``vamb.parsecontigs.Composition`` (SURVEY.md section 8f,
row N2): ``FastaEntry.kmercounts`` (``vamb/vambtools.py:444-447``, Rust ``vambcore.kmercounts`` in the reference) and
``Composition._project`` (``vamb/parsecontigs.py:140-150``) followed by ``mask_lower_bits(., 12)`` (``parsecontigs.py:211``).

The FASTA parser, name handling and the ``Composition`` container stay the reference's; this module offers the two
numerical steps batched over many contigs (one upload of the concatenated sequences, counts never leave the device):

    tnf = TnfProjector(kernel).from_sequences([b"ACGT...", ...])          # [n, 103] float32, low 12 bits cleared
    proj = TnfProjector(kernel).project(fourmers)                          # drop-in body of Composition._project

``kernel`` is the reference's ``vamb/kernel.npz`` (float32 [256, 103]); inside Vamb it is ``vamb.parsecontigs._KERNEL``.
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, Sequence

import numpy as _np

from . import _lib

NKMERS = 256
NTNF = 103


class TnfProjector:
    def __init__(self, kernel: _np.ndarray):
        kernel = _np.ascontiguousarray(kernel, dtype=_np.float32)
        if kernel.shape != (NKMERS, NTNF):
            raise ValueError(f"projection kernel must be of shape (256, 103), not {kernel.shape}")
        self._lib = _lib.load()
        _lib.require_gpu()
        h = ctypes.c_void_p()
        _lib.check(self._lib.vh_tnf_create(_lib.ptr(kernel), ctypes.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            self._lib.vh_tnf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _concat(sequences: Sequence[bytes]):
        offsets = _np.zeros(len(sequences) + 1, dtype=_np.int64)
        _np.cumsum([len(s) for s in sequences], out=offsets[1:])
        bases = _np.frombuffer(b"".join(bytes(s) for s in sequences), dtype=_np.uint8)
        return _np.ascontiguousarray(bases), offsets

    def kmercounts(self, sequences: Sequence[bytes]) -> _np.ndarray:
        """uint32 [n, 256]: ``FastaEntry(...).kmercounts()`` of every sequence (vambtools.py:444-447)."""
        bases, offsets = self._concat(sequences)
        counts = _np.zeros((len(sequences), NKMERS), dtype=_np.uint32)
        _lib.check(self._lib.vh_tnf_kmercounts(self._h, _lib.ptr(bases) if len(bases) else None, _lib.ptr(offsets),
                                               len(sequences), _lib.ptr(counts)))
        return counts

    def project(self, fourmers: _np.ndarray, mask_bits: int = 0) -> _np.ndarray:
        """``Composition._project(fourmers, kernel)`` (parsecontigs.py:140-150) for float32 [n, 256] raw counts; the input is
        not modified (the reference normalises it in place).  ``mask_bits=12`` adds ``mask_lower_bits`` (parsecontigs.py:211)."""
        fourmers = _np.ascontiguousarray(fourmers, dtype=_np.float32)
        if fourmers.ndim != 2 or fourmers.shape[1] != NKMERS:
            raise ValueError("fourmers must be of shape (n, 256)")
        out = _np.empty((len(fourmers), NTNF), dtype=_np.float32)
        _lib.check(self._lib.vh_tnf_project(self._h, _lib.ptr(fourmers), len(fourmers), int(mask_bits), _lib.ptr(out)))
        return out

    def from_sequences(self, sequences: Sequence[bytes], mask_bits: int = 12) -> _np.ndarray:
        """Counts and projection without the counts leaving the device: the TNF matrix ``Composition.from_file`` builds
        (parsecontigs.py:184-211) for the sequences that passed its length filter.  Raises the reference's ValueError for a
        sequence without a single countable 4-mer."""
        bases, offsets = self._concat(sequences)
        n = len(sequences)
        counts = _np.zeros((n, NKMERS), dtype=_np.uint32)
        _lib.check(self._lib.vh_tnf_kmercounts(self._h, _lib.ptr(bases) if len(bases) else None, _lib.ptr(offsets), n,
                                               _lib.ptr(counts)))
        empty = _np.flatnonzero(counts.sum(axis=1) == 0)
        if len(empty):
            raise ValueError(f"TNF value of contig number {int(empty[0])} is all zeros. This implies that the sequence "
                             "contained no 4-mers of A, C, G, T, making this sequence uninformative.")
        out = _np.empty((n, NTNF), dtype=_np.float32)
        _lib.check(self._lib.vh_tnf_project(self._h, None, n, int(mask_bits), _lib.ptr(out)))
        return out
