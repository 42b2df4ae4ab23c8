#!/bin/bash
# Round 3, GPU call 6: fused publication of scan passes -- golden streams, then the C2 sweep A/B on ONE trained latent matrix
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_cluster_gpu.py -m gpu -x -q > $O/pytest_cluster.log 2>&1; echo "rc=$?" >> $O/pytest_cluster.log); tail -6 $O/pytest_cluster.log
timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_SCAN_FUSED_PUBLISH=1,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SCAN_FUSED_PUBLISH=0,VAMBHIP_GEN_PROFILE=1;VAMBHIP_SCAN_FUSED_PUBLISH=1" $O/sweep_ab.json 2> $O/sweep_ab.err | tee $O/sweep_ab.txt
grep vambhip $O/sweep_ab.err | head -60
