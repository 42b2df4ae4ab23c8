#!/bin/bash
# round 3, call 38: reference-order mode (scan.reference_order): normalisation / accumulators vs the oracle in that order, streams
# vs the REAL reference's goldens (all fixtures, both 100 k ones in full); then the whole cluster file (default order unchanged)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zk
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_cluster_gpu.py -m gpu -q -x -k "reference_order" > $O/pytest_reference_order.log 2>&1; tail -4 $O/pytest_reference_order.log
timeout 600 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py -m gpu -q -x -k "not reference_order" > $O/pytest_cluster_default.log 2>&1; tail -2 $O/pytest_cluster_default.log
