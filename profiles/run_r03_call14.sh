#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
for args in "bf16 3 steps 1 0.2" "bf16 3 steps 9 0.2" "bf16 3 epochs 9 0.2" "bf16 3 epochs 9 0.0"; do
  timeout 200 python tools/gpu/gpu_determinism.py $args 2>&1 | grep -v amdgpu | tee -a $O/determinism_matrix.txt
done
(timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); grep -v "INFO " $O/pytest_gpu.log | tail -12
