#!/bin/bash
# Round 3, GPU call 8: neighbourhood speculation of the native cluster state machine -- golden streams, then the C2 sweep A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_cluster_gpu.py -m gpu -x -q > $O/pytest_cluster.log 2>&1; echo "rc=$?" >> $O/pytest_cluster.log); tail -6 $O/pytest_cluster.log
timeout 1500 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_NEIGHBOURS=0,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_WINDOW=16,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SPEC_WINDOW=4,VAMBHIP_GEN_PROFILE=1" $O/sweep_ab.json 2> $O/sweep_ab.err | tee $O/sweep_ab.txt
grep "vambhip\] generator\|passes by purpose" $O/sweep_ab.err | cut -c1-700
