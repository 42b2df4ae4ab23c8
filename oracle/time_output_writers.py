#!/usr/bin/env python
"""Build container only: wall time of the reference's cluster_and_write_files (oracle/ref_output.py) and of
vamb_amd.output.cluster_and_write_files on the same synthetic stream of clusters (no clustering: a canned generator).
    python oracle/time_output_writers.py [n_contigs] [mean cluster size]"""
import os, sys, tempfile, time, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(ROOT, "tests", "golden")]
import ref_output  # noqa: E402
from vamb_amd import output  # noqa: E402
from vamb_amd.cluster import Cluster  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.RandomState(0)
perm = rng.permutation(n)
cuts = np.sort(rng.choice(np.arange(1, n), size=n // mean - 1, replace=False))
groups = np.split(perm, cuts)
names = [f"S{i % 7}C{i}" for i in range(n)]
lens = rng.randint(2000, 50000, size=n)
latent = np.zeros((n, 1), np.float32)


def canned(*a, **k):
    for g in groups:
        yield Cluster(int(g[0]), int(g[0]), np.sort(g), 0.1, 0.3, 0.07, 1, 1)


ref_fn, vt = ref_output.load_cluster_and_write_files(canned)
opts = types.SimpleNamespace(window_size=300, min_successes=15, max_clusters=None)
for tag, fn, kw in (("reference", ref_fn, {}), ("vamb_amd.output", output.cluster_and_write_files, dict(_cluster_generator=canned))):
    sp = vt.BinSplitter("C"); sp.initialize(names[:1000])
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        fn(opts, sp, latent, names, lens, 0, False, os.path.join(tmp, "x"), None, None, **kw)
        print(f"{tag}: {time.perf_counter() - t0:.2f} s for {n} contigs in {len(groups)} clusters (split by sample)")
