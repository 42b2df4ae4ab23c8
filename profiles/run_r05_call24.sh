#!/bin/bash
# Round 5, GPU call 24: speculative fill one pass ahead (gen.prefill): cluster / parallel tests, C2 sweep A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py -m gpu -q --maxfail=8 > $O/pytest_cluster.log 2>&1; grep -E "passed|failed" $O/pytest_cluster.log
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PREFILL=1;VAMBHIP_GEN_PREFILL=0;VAMBHIP_GEN_PREFILL=1" $O/sweep_prefill.json > $O/sweep_prefill.txt 2>&1; grep -v "passes with" $O/sweep_prefill.txt | grep -v amdgpu.ids | cut -c1-420
