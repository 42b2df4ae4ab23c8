#!/bin/bash
# round 3, call 32: speculative bookkeeping deferred under the next pass (gen.defer_bookkeeping) -- tests + C2 sweep A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ze
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > $O/pytest_cluster.log 2>&1; tail -2 $O/pytest_cluster.log
timeout 1500 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1;VAMBHIP_DEFER_BOOKKEEPING=0,VAMBHIP_GEN_PROFILE=1;VAMBHIP_GEN_PROFILE=1" $O/sweep_ab.json 2> $O/sweep_ab.err | tee $O/sweep_ab.txt
grep "vambhip\] generator\|host time" $O/sweep_ab.err | cut -c1-400
