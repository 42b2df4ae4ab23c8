#!/bin/bash
# Round 6, GPU call 11: the pass published by the scan kernel's own last workgroup (scan.fused_publish): cluster / parallel / e2e /
# determinism tests, C2 sweep A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py tests/test_determinism_gpu.py tests/test_cli_gpu.py -m gpu -q --maxfail=8 > $O/pytest_cluster.log 2>&1; tail -6 $O/pytest_cluster.log | cut -c1-250
S="VAMBHIP_SCAN_FUSED_PUBLISH"
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=1;$S=0;$S=1;$S=0" $O/sweep_fused_publish.json > $O/sweep_fused_publish.txt 2>&1; grep -v "passes with" $O/sweep_fused_publish.txt | grep -v amdgpu.ids | grep "setting\|generator: total" | cut -c1-260
