#!/bin/bash
# Round 6, GPU call 15: tile of the elementwise BatchNorm-backward kernel (vae_dz16_kernel<COLS, ROWS>), step A/B at C2 and the C3 shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
Z="VAMBHIP_VAE_DZ_TILE"
timeout 1200 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$Z=1|$Z=2|$Z=3|$Z=4|$Z=5|$Z=6|$Z=7" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$Z=1|$Z=3|$Z=4|$Z=5" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
