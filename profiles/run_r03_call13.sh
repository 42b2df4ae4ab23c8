#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
for args in "bf16 3 steps 1 0.2" "bf16 3 steps 2 0.2" "bf16 3 steps 9 0.2" "bf16 3 epochs 1 0.2" "bf16 3 epochs 2 0.2" "bf16 3 epochs 9 0.2" "bf16 3 epochs 9 0.0" "bf16 3 steps 9 0.0" "fp32 3 epochs 9 0.2"; do
  timeout 200 python tools/gpu/gpu_determinism.py $args 2>&1 | grep -v amdgpu | tee -a $O/determinism_matrix.txt
done
