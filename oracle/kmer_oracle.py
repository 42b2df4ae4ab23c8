"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the TNF compute.

``kmercounts``: the definition the reference's own test checks ``vambcore.kmercounts`` against
(``test/test_vambtools.py:137-151``): every window of 4 bytes, upper-cased, that is a word over ACGT counts at the index
of that word in ``itertools.product("ACGT", repeat=4)``; every other byte voids its windows.  (Whether the Rust code counts
U is UNPINNED -- its source is not in the reference tree and the error text at ``vamb/parsecontigs.py:194-199`` names U --
the restatement follows the pinned definition, which does not.)
``project``: the statements of ``Composition._project`` (``vamb/parsecontigs.py:140-150``) in numpy.
Pinned by ``tests/golden/tnf_case.npz``, whose TNF rows come from the real ``Composition._project``."""
from __future__ import annotations

import numpy as np

_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c | 0x20] = _i


def kmercounts(seq: bytes) -> np.ndarray:
    b = _CODE[np.frombuffer(bytes(seq), dtype=np.uint8)]
    out = np.zeros(256, dtype=np.uint32)
    if len(b) < 4:
        return out
    w = np.stack([b[0:-3], b[1:-2], b[2:-1], b[3:]])
    ok = (w < 4).all(axis=0)
    idx = (w[0].astype(np.int64) << 6) | (w[1].astype(np.int64) << 4) | (w[2].astype(np.int64) << 2) | w[3]
    np.add.at(out, idx[ok], 1)
    return out


def project(fourmers: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    fourmers = np.array(fourmers, dtype=np.float32, copy=True)
    s = fourmers.sum(axis=1).reshape(-1, 1)
    s[s == 0] = 1.0
    fourmers *= 1 / s
    fourmers += -(1 / 256)
    return np.dot(fourmers, kernel)
