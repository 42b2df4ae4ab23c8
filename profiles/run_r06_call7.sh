#!/bin/bash
# Round 6, GPU call 7: new step defaults (paired weight gradients, fork plan by input width, optimiser tail behind two-level tickets)
# + removed rows riding in the next scan's kernel arguments (gen.inline_removals): whole GPU suite, C2 sweep A/B, step check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log | cut -c1-250
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_INLINE_REMOVALS=1;VAMBHIP_GEN_INLINE_REMOVALS=0;VAMBHIP_GEN_INLINE_REMOVALS=1;VAMBHIP_GEN_INLINE_REMOVALS=0" $O/sweep_inline_rm.json > $O/sweep_inline_rm.txt 2>&1; grep -v "passes with" $O/sweep_inline_rm.txt | grep -v amdgpu.ids | cut -c1-330
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_FORK_PLAN=6;VAMBHIP_VAE_FUSED_FINALIZE=0;VAMBHIP_VAE_DW_PAIR=0" 2 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|VAMBHIP_VAE_FORK_PLAN=6;VAMBHIP_VAE_FUSED_FINALIZE=0;VAMBHIP_VAE_DW_PAIR=0" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
