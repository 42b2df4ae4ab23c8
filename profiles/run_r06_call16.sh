#!/bin/bash
# Round 6, GPU call 16: narrower / taller tiles of vae_dz16_kernel (fewer fp64 atomics per column)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
Z="VAMBHIP_VAE_DZ_TILE"
timeout 1200 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$Z=7|$Z=8|$Z=9|$Z=10|$Z=11|$Z=12|$Z=13" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$Z=7|$Z=8|$Z=9|$Z=10|$Z=12" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
