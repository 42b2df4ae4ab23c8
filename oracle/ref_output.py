"""TEST INFRASTRUCTURE ONLY -- runs the reference's OWN ``cluster_and_write_files`` (``vamb/__main__.py:1254-1404``).

``vamb/__main__.py`` cannot be imported in this image (it pulls in pycoverm / pyhmmer / pyrodigal at module level), so the
function's source segment -- and ``ceil_div`` (``__main__.py:1181-1182``) -- is cut out of the file with ``ast`` and executed
unmodified in a namespace that provides the names it refers to: the real ``vamb.vambtools`` / ``vamb.cluster`` modules of
``ref_harness``, a null logger, and the standard-library names ``__main__`` imports.  Build container only."""
from __future__ import annotations

import ast
import itertools
import os
import sys
import time
from contextlib import nullcontext
from typing import Optional, Sequence, cast

import numpy as np

import ref_harness


def load_cluster_and_write_files(cluster_generator=None):
    """The reference function; ``cluster_generator`` (if given) replaces ``vamb.cluster.ClusterGenerator`` inside it."""
    vt, cl, _ = ref_harness.load_reference()
    path = os.path.join(ref_harness.REFERENCE_ROOT, "vamb", "__main__.py")
    source = open(path).read()
    tree = ast.parse(source)
    wanted = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("cluster_and_write_files", "ceil_div"):
            wanted[node.name] = ast.get_source_segment(source, node)
    assert set(wanted) == {"cluster_and_write_files", "ceil_div"}

    class _Cluster:
        ClusterGenerator = cl.ClusterGenerator if cluster_generator is None else cluster_generator

    class _Vamb:
        vambtools = vt
        cluster = _Cluster

    ns = dict(vamb=_Vamb, np=np, itertools=itertools, time=time, nullcontext=nullcontext, Optional=Optional,
              Sequence=Sequence, cast=cast, logger=ref_harness._NullLogger(), ClusterOptions=object, FastaOutput=object,
              create_cluster_fasta_files=None)
    exec(wanted["ceil_div"], ns)
    exec(wanted["cluster_and_write_files"], ns)
    return ns["cluster_and_write_files"], vt
