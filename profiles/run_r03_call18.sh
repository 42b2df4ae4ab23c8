#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
timeout 1500 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_DEPTH=1,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_WINDOW=16,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_WINDOW=4,VAMBHIP_GEN_PROFILE=1" $O/sweep_ab.json 2> $O/sweep_ab.err | tee $O/sweep_ab.txt
grep "vambhip\] generator\|passes by purpose\|host time" $O/sweep_ab.err | cut -c1-900
