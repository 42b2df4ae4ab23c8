#!/bin/bash
# Round 3, GPU call 7: what the passes of a C2 sweep are FOR (generator profile) + run-to-run determinism, unfused vs fused publish
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout 1200 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1;A=1;VAMBHIP_SCAN_FUSED_PUBLISH=1;VAMBHIP_SCAN_FUSED_PUBLISH=1;B=1" $O/sweep_ab.json 2> $O/sweep_ab.err | tee $O/sweep_ab.txt
grep vambhip $O/sweep_ab.err | head -8
