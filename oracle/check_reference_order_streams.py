#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (oracle/): the restated state machine with distances / normalisation in the REFERENCE's evaluation
order (cluster_oracle.set_order(1)) against the real reference's golden streams of the two 100 k fixtures (minutes of CPU).

    python oracle/check_reference_order_streams.py > profiles/r03_reference_order_streams.txt
"""
import sys, time, numpy as np
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))
import cluster_oracle as co, fixture_defs as fd
for order in (1, 2, 0):
    co.set_order(order)
    for name in fd.CLUSTER_CASES_LARGE:
        mat, lens, kw = fd.cluster_inputs(name)
        t0 = time.time()
        packed = fd.pack_stream(list(co.OracleClusterGenerator(mat.copy(), lens, **kw)))
        ref = fd.load("cluster_" + name)
        n = min(len(packed["medoid"]), len(ref["medoid"]))
        diff = np.flatnonzero(packed["medoid"][:n] != ref["medoid"][:n])
        ok, msg = fd.streams_equal(packed, ref, pvr_rtol=0.0 if order == 1 else 1e-2)   # reference order: EVERY field exact
        print(f"order {order} {name:24s} clusters {len(packed['medoid']):6d} (reference {len(ref['medoid']):6d}) identical prefix {int(diff[0]) if len(diff) else n:6d} equal={ok} {'' if ok else msg[:100]}  [{time.time()-t0:.0f}s]", flush=True)
co.set_order(co.DEFAULT_ORDER)
