"""CPU: the TNF restatement (oracle/kmer_oracle.py) against the golden case generated from the real reference
(tests/golden/make_golden.py tnf: kernel.npz and Composition._project from vamb/parsecontigs.py; the counts by the definition
test/test_vambtools.py:137-151 pins vambcore.kmercounts to)."""
import numpy as np

import fixture_defs as fd
import kmer_oracle as ko


def test_kmercounts_restatement_matches_pinned_definition():
    g = fd.load("tnf_case")
    seqs = fd.tnf_sequences()
    got = np.stack([ko.kmercounts(s) for s in seqs])
    assert got.dtype == np.uint32 and np.array_equal(got, g["counts"])
    assert got[-2].sum() == 0 and got[-1].sum() == 0          # shorter than a 4-mer / no countable 4-mer
    # lower case counts, ambiguity codes void the four windows they are part of
    assert ko.kmercounts(b"acgtACGT").sum() == 5 and ko.kmercounts(b"ACGNACGT").sum() == 1


def test_projection_restatement_matches_reference():
    g = fd.load("tnf_case")
    p = ko.project(g["counts"].astype(np.float32), g["kernel"])
    assert np.array_equal(p, g["projected"])          # same numpy statements on the same machine
    assert g["kernel"].shape == (256, 103) and (g["tnf"].view(np.uint32) & 0xFFF).max() == 0
