#!/bin/bash
# round 3, call 41: scan.reference_order = 2 (tuned kernels as a filter, ref_dot in the drain) beside mode 1, then the
# default-order cluster tests on the same build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zn
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_cluster_gpu.py -m gpu -q -k "reference_order" > $O/pytest_reference_order.log 2>&1; tail -6 $O/pytest_reference_order.log | cut -c1-200
timeout 200 python -m pytest tests/test_cluster_gpu.py -m gpu -q -x -k "not reference_order" > $O/pytest_cluster_default.log 2>&1; tail -2 $O/pytest_cluster_default.log
