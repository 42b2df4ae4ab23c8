#!/bin/bash
# Round 5, GPU call 16: the loss kernel reads its targets from the dataset rows of the batch (no fp32 copy of the batch): tests, A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_dp_gpu.py tests/test_e2e_gpu.py tests/test_parallel_gpu.py tests/test_semisup_gpu.py -m gpu -q --maxfail=8 > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log
timeout 300 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_LOSS_FROM_DATASET=0" 2 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|VAMBHIP_VAE_LOSS_FROM_DATASET=0|VAMBHIP_VAE_PREFETCH_MAX_COLS=2048" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
