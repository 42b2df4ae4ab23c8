"""The output side of ``vamb bin default`` -- ``cluster_and_write_files`` (``vamb/__main__.py:1254-1404``) with the reference's
signature, vectorised.

The reference prints every (cluster, contig) pair with its own ``print`` call and sums the cluster's base pairs with a Python
generator over numpy scalars; with the sweep on the GPU this loop is what remains of the wall clock at millions of contigs.
Here every cluster is one ``write`` per file (lines joined in memory) and one numpy reduction.  The three files
(``*_metadata.tsv``, ``*_unsplit.tsv``, ``*_split.tsv``) have the reference's content; the lines of one split bin come in
member order here and in ``set`` iteration order (hash-seed dependent) in the reference (``vambtools.py:115-141``).
``tests/test_output_host.py`` runs the REAL reference function (``oracle/ref_output.py``) beside this one on the same stream.
"""
from __future__ import annotations

import itertools
import logging
import time
from contextlib import nullcontext
from typing import Optional, Sequence

import numpy as _np

try:
    from loguru import logger
except ImportError:   # pragma: no cover
    logger = logging.getLogger("vamb_amd.output")

CLUSTERS_HEADER = "clustername\tcontigname"          # vamb/vambtools.py:18
METADATA_HEADER = "name\tradius\tpeak valley ratio\tkind\tbp\tncontigs\tmedoid"   # vamb/__main__.py:1312-1316


def _ceil_div(num: int, den: int) -> int:
    return -(num // -den)


def _split_members(binsplitter, cluster_name: str, members: list):
    """``BinSplitter.split_bin`` (vambtools.py:115-141) with lists in member order instead of sets."""
    splitter = binsplitter.splitter
    by_sample: dict = {}
    for identifier in members:
        sample, _, rest = identifier.partition(splitter)
        if not rest or not sample:
            raise KeyError(f"Separator '{splitter}' not in sequence identifier, or is at the very start or end of "
                           f"identifier: '{identifier}'")
        by_sample.setdefault(sample, []).append(identifier)
    for sample, headers in by_sample.items():
        yield f"{sample}{splitter}{cluster_name}", list(dict.fromkeys(headers))   # a set in the reference: no duplicates


def cluster_and_write_files(cluster_options, binsplitter, latent: _np.ndarray, sequence_names: Sequence[str],
                            sequence_lens: _np.ndarray, seed: int, cuda: bool, base_clusters_name: str,
                            fasta_output, bin_prefix: Optional[str], _cluster_generator=None,
                            _create_cluster_fasta_files=None):
    begintime = time.time()
    logger.info("Clustering")
    logger.info(f"\tWindowsize: {cluster_options.window_size}")
    logger.info(f"\tMin successful thresholds detected: {cluster_options.min_successes}")
    logger.info(f"\tMax clusters: {cluster_options.max_clusters}")
    logger.info(f"\tUse CUDA for clustering: {cuda}")
    logger.info(f"\tBinsplitter: {binsplitter.log_string()}")

    if _cluster_generator is None:
        from .cluster import ClusterGenerator as _cluster_generator
    cluster_generator = _cluster_generator(latent, sequence_lens, windowsize=cluster_options.window_size,
                                           minsuccesses=cluster_options.min_successes, destroy=True, normalized=False,
                                           cuda=cuda, rng_seed=seed)
    clusters = itertools.islice(cluster_generator, cluster_options.max_clusters)
    split = not binsplitter.is_disabled()
    context = open(base_clusters_name + "_split.tsv", "w") if split else nullcontext(None)
    stored = None if fasta_output is None else []
    names = _np.asarray(sequence_names, dtype=object)
    lens = _np.asarray(sequence_lens)
    n_processed = n_split = n_unsplit = 0
    with open(base_clusters_name + "_metadata.tsv", "w") as metadata_file, \
            open(base_clusters_name + "_unsplit.tsv", "w") as unsplit_file, context as split_file:
        metadata_file.write(METADATA_HEADER + "\n")
        unsplit_file.write(CLUSTERS_HEADER + "\n")
        if split_file is not None:
            split_file.write(CLUSTERS_HEADER + "\n")
        n_total = latent.shape[0]
        last_decile = 0
        for cluster_index, cluster in enumerate(clusters):
            idx = _np.asarray(cluster.members, dtype=_np.int64)
            members = names[idx].tolist()
            cluster_name = str(cluster_index + 1)
            if bin_prefix is not None:
                cluster_name = bin_prefix + cluster_name
            n_processed += len(members)
            n_unsplit += 1
            prefix = cluster_name + "\t"
            unsplit_file.write(prefix + ("\n" + prefix).join(members) + "\n" if members else "")
            if split_file is None:
                if stored is not None:
                    stored.append((cluster_name, members))
            else:
                for split_name, split_members in _split_members(binsplitter, cluster_name, members):
                    n_split += 1
                    if stored is not None:
                        stored.append((split_name, split_members))
                    sp = split_name + "\t"
                    split_file.write(sp + ("\n" + sp).join(split_members) + "\n")
            radius = None if cluster.radius is None else round(cluster.radius, 3)
            pvr = None if cluster.observed_pvr is None else round(cluster.observed_pvr, 2)
            bp = int(lens[idx].sum()) if len(idx) else 0
            metadata_file.write(f"{cluster_name}\t{radius}\t{pvr}\t{cluster.kind_str}\t{bp}\t{len(members)}\t"
                                f"{names[cluster.medoid]}\n")
            current_decile = _ceil_div(10 * n_processed, n_total)
            for decile in range(last_decile + 1, current_decile + 1):
                logger.info(f"\t {decile * 10:3} % of contigs clustered")
            last_decile = current_decile

    binsplitter.log_clustering_result(n_total, n_split, n_unsplit, begintime)
    if fasta_output is not None:
        if _create_cluster_fasta_files is None:
            from vamb.__main__ import create_cluster_fasta_files as _create_cluster_fasta_files
        _create_cluster_fasta_files(fasta_output.bins_dir_to_populate, stored, fasta_output.existing_fasta_path.path,
                                    sequence_lens, sequence_names, fasta_output.min_fasta_size,
                                    fasta_output.compress_output)
