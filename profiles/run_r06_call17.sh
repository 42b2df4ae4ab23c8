#!/bin/bash
# Round 6, GPU call 17: VAE tests on the 64 x 128 dz16 tile; split-K width of the weight gradients (vae.dw_workgroups) at the step level
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_semisup_gpu.py tests/test_determinism_gpu.py -m gpu -q -x > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log | cut -c1-300
W="VAMBHIP_DW_WGS"
timeout 1200 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$W=128|$W=192|$W=384|$W=512" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$W=128|$W=192|$W=384|$W=512" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
