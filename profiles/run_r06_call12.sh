#!/bin/bash
# Round 6, GPU call 12: two-workgroups-per-CU tiles of the training GEMM; reported observed_pvr vs the reference's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
timeout 600 python tools/gpu/gpu_gemm16_twowg.py $O/gemm16_twowg.json > $O/gemm16_twowg.txt 2>&1; grep -v amdgpu.ids $O/gemm16_twowg.txt | tail -40
timeout 900 python tools/gpu/gpu_pvr_deviation.py > $O/pvr_deviation.txt 2>&1; grep -v amdgpu.ids $O/pvr_deviation.txt | tail -20
