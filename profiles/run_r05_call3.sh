#!/bin/bash
# Round 5, GPU call 3: the matrix-pipe scan kernel with unconditional prefetches (counted vmcnt instead of a drain per tile),
# the two-stream schedule variants of the step, the new reference-recorded C1-shape golden, the joint-trainer binding.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_vae_gpu.py tests/test_cluster_gpu.py -m gpu -q --maxfail=8 > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
timeout 300 python tools/gpu/gpu_scan_dbg.py 620000 32 $O/scan_dbg_k32.txt > /dev/null 2>&1; cat $O/scan_dbg_k32.txt
timeout 300 python tools/gpu/gpu_scan_dbg.py 170000 16 $O/scan_dbg_k16_small.txt > /dev/null 2>&1; head -4 $O/scan_dbg_k16_small.txt
P="VAMBHIP_VAE_FORK_PLAN"; M="VAMBHIP_VAE_FORK_MODE"
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 200 8192 10 bf16 "|$P=1|$P=2|$P=4|$P=3|$P=6|$P=7|VAMBHIP_VAE_FORK_AT_LOSS=1|VAMBHIP_VAE_FORK_AT_LOSS=1;$P=6|$M=2|$M=2;$P=7|$M=2;$P=7;VAMBHIP_VAE_FORK_AT_LOSS=1" 2 > $O/step_fork_plans_c2.txt 2>&1; grep SUMMARY $O/step_fork_plans_c2.txt
VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1" $O/sweep.json > $O/sweep.txt 2>&1; grep -v "passes with" $O/sweep.txt | tail -6; grep "passes with 32\|passes with 16\|passes with  8\|passes with  1 " $O/sweep.txt
