#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); grep -v "INFO " $O/pytest_gpu.log | tail -40
