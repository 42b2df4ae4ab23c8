#!/bin/bash
# Round 6, GPU call 14: fold-in-GEMM with the butterflies side by side and the output layer restricted to <= 512 columns; the narrow-K tile alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -x -k "fold_in_gemm or dropout_statistics or c2_shape or c1_shape" > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log | cut -c1-300
F="VAMBHIP_VAE_FOLD_IN_GEMM"; T="VAMBHIP_VAE_NARROW_K_TILE"
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$F=0|$F=0;$T=0|$T=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$F=0" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
timeout 900 python tools/gpu/gpu_pvr_deviation.py > $O/pvr_deviation.txt 2>&1; grep -v amdgpu.ids $O/pvr_deviation.txt | tail -12
