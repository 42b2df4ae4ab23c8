#!/bin/bash
# Round 6, GPU call 32: where the host's ~2.2 s per C2 sweep go -- inclusive timers of the pieces of an emission (gen.profile)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y8; mkdir -p $O; cd $R
VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1;X=1" > $O/sweep_profile.txt 2>&1
grep "setting\|generator: total\|host time\|inclusive\|passes by purpose" $O/sweep_profile.txt | cut -c1-700
