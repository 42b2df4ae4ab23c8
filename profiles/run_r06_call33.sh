#!/bin/bash
# Round 6, GPU call 33: up to how many resident rows is a pass with <= 8 needed medoids worth widening to 32 slots (gen.widen_max_rows;
# 600 k since round 4, before the row-major kernel and the two-step publish made the 32-slot pass cheaper)?  C2 sweeps, one process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y9; mkdir -p $O; cd $R
S=VAMBHIP_GEN_WIDEN_MAX_ROWS
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=600000;$S=900000;$S=1300000;$S=2100000;$S=400000;$S=600000;$S=900000;$S=1300000" > $O/sweep_widen.txt 2>&1
grep "setting\|generator: total" $O/sweep_widen.txt | cut -c1-260
