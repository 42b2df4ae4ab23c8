#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout 300 python tests/diagnostics/gpu_determinism_step.py bf16 4 512 6 2>&1 | grep -v amdgpu | tee $O/determinism_step_bf16.txt
echo "--- round-2 dataflow"; VAMBHIP_VAE_GEMM_PIPELINE=0 VAMBHIP_VAE_DW_ROW_MAJOR=0 timeout 300 python tests/diagnostics/gpu_determinism_step.py bf16 3 512 6 2>&1 | grep -v amdgpu | tee $O/determinism_step_bf16_r2.txt
echo "--- fp32"; timeout 300 python tests/diagnostics/gpu_determinism_step.py fp32 3 512 6 2>&1 | grep -v amdgpu | tail -8
