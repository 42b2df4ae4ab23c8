#!/bin/bash
# Round 6, GPU call 5: where the paired weight-gradient launch pays: fork plans x vae.dw_pair, the optimiser's tail behind two-level
# arrival tickets (vae.fused_finalize), at C2 and the C3 shape; scheduling-variant test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -q --maxfail=8 -k "scheduling_variants or fused" > $O/pytest_vae.log 2>&1; tail -5 $O/pytest_vae.log | cut -c1-300
P="VAMBHIP_VAE_DW_PAIR"; F="VAMBHIP_VAE_FORK_PLAN"; Z="VAMBHIP_VAE_FUSED_FINALIZE"
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$P=0|$F=14|$F=14;$P=0|$F=10|$F=2|$F=12|$Z=1|$Z=1;$F=14|$F=14;VAMBHIP_VAE_FORK_AT_LOSS=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$P=0|$F=14|$Z=1|$Z=1;$F=14" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
