#!/bin/bash
# Round 5, GPU call 25: fill one pass ahead with a bounded fresh walk in front (gen.prefill = 2) against the plain list (1) and none (0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PREFILL=1;VAMBHIP_GEN_PREFILL=2;VAMBHIP_GEN_PREFILL=2,VAMBHIP_GEN_PREFILL_FRESH_SEEDS=16;VAMBHIP_GEN_PREFILL=0;VAMBHIP_GEN_PREFILL=1,VAMBHIP_GEN_PREFILL_FRESH_SEEDS=4" $O/sweep_prefill2.json > $O/sweep_prefill2.txt 2>&1; grep -v "passes with\|passes by purpose" $O/sweep_prefill2.txt | grep -v amdgpu.ids | cut -c1-330
