#!/bin/bash
# Round 5, GPU call 6: the row-major matrix-pipe scan pass (K6r): bit-exact cluster streams, pass time, a C2 sweep A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_parallel_gpu.py tests/test_dp_gpu.py tests/test_e2e_gpu.py -m gpu -q --maxfail=8 > $O/pytest_cluster.log 2>&1; tail -4 $O/pytest_cluster.log
SCAN_DBG_LIST=0,13,1,8,4 timeout 300 python tools/gpu/gpu_scan_dbg.py 620000 32 $O/scan_dbg_k32_rm.txt > /dev/null 2>&1; cat $O/scan_dbg_k32_rm.txt
SCAN_DBG_LIST=0,13 VAMBHIP_SCAN_MFMA_ROWMAJOR=0 timeout 300 python tools/gpu/gpu_scan_dbg.py 620000 32 $O/scan_dbg_k32_cm.txt > /dev/null 2>&1; cat $O/scan_dbg_k32_cm.txt
SCAN_DBG_LIST=0,13 timeout 300 python tools/gpu/gpu_scan_dbg.py 170000 16 $O/scan_dbg_k16_rm.txt > /dev/null 2>&1; cat $O/scan_dbg_k16_rm.txt
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_SCAN_MFMA_ROWMAJOR=1;VAMBHIP_SCAN_MFMA_ROWMAJOR=0;VAMBHIP_SCAN_MFMA_ROWMAJOR=1" $O/sweep_ab.json > $O/sweep_ab.txt 2>&1; grep "cluster_s\|generator: total\|passes with 32\|passes with 16 \|passes with  8 \|passes with 24" $O/sweep_ab.txt
