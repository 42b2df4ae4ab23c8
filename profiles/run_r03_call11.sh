#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
timeout 600 python tools/gpu/gpu_determinism.py bf16 3 "A=0;VAMBHIP_SINGLE_STREAM=1;VAMBHIP_VAE_GEMM_PIPELINE=0;VAMBHIP_VAE_DW_ROW_MAJOR=0;VAMBHIP_VAE_GEMM_PIPELINE=0,VAMBHIP_VAE_DW_ROW_MAJOR=0;VAMBHIP_VAE_GEMM_PIPELINE=0,VAMBHIP_VAE_DW_ROW_MAJOR=0,VAMBHIP_SINGLE_STREAM=1;VAMBHIP_SINGLE_STREAM=1,VAMBHIP_VAE_GEMM_PIPELINE=0" 2>&1 | grep -v amdgpu | tee $O/determinism_bf16.txt
DET_EPOCHS=1 timeout 600 python tools/gpu/gpu_determinism.py bf16 3 "A=0;VAMBHIP_SINGLE_STREAM=1" 2>&1 | grep -v amdgpu | tee -a $O/determinism_bf16.txt
