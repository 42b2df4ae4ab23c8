#!/bin/bash
# Round 3, GPU call 5: training step time A/B without a profiler (400 k contigs x 200 samples, batch 8192, bf16: 48 steps / epoch)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
for opt in "A=default" "VAMBHIP_SINGLE_STREAM=1" "VAMBHIP_FORK_EVENTS=1" "VAMBHIP_VAE_GEMM_PIPELINE=0" "VAMBHIP_VAE_DW_ROW_MAJOR=0" "VAMBHIP_VAE_GEMM_PIPELINE=0 VAMBHIP_VAE_DW_ROW_MAJOR=0"; do
  for rep in 1 2; do
    echo -n "[$opt] " | tee -a $O/step_time_ab.txt
    env $opt timeout 300 python tools/gpu/gpu_epoch_time.py 400000 200 8192 40 bf16 2>/dev/null | tee -a $O/step_time_ab.txt
  done
done
