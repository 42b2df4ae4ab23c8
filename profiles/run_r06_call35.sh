#!/bin/bash
# Round 6, GPU call 35: the look-ahead window of the speculative fill (gen.spec_window, 16 since round 4) re-measured with the cheaper
# 32-slot pass of this round: C2 sweeps in one process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y11; mkdir -p $O; cd $R
S=VAMBHIP_SPEC_WINDOW
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=16;$S=24;$S=32;$S=12;$S=16;$S=24;$S=32;$S=12" > $O/sweep_spec_window.txt 2>&1
grep "setting\|generator: total" $O/sweep_spec_window.txt | cut -c1-260
