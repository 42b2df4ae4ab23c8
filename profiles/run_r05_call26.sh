#!/bin/bash
# Round 5, GPU call 26: fill one pass ahead after the ring fix: determinism of the sweep's counters (two runs each), tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PREFILL=2;VAMBHIP_GEN_PREFILL=0;VAMBHIP_GEN_PREFILL=2;VAMBHIP_GEN_PREFILL=1;VAMBHIP_GEN_PREFILL=1" $O/sweep_prefill3.json > $O/sweep_prefill3.txt 2>&1; grep -v "passes with" $O/sweep_prefill3.txt | grep -v amdgpu.ids | cut -c1-700 | grep -v "host time inside"
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py -m gpu -q --maxfail=8 > $O/pytest_cluster.log 2>&1; grep -E "passed|failed" $O/pytest_cluster.log
