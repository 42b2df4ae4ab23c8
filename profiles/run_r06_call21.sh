#!/bin/bash
# Round 6, GPU call 21: loss kernel with the rows in registers: parity with the staged kernel, VAE tests, step A/B at C2 and the C3 shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py -m gpu -q > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log | cut -c1-300
L="VAMBHIP_VAE_LOSS_REGISTERS"
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$L=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$L=0" 3 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
