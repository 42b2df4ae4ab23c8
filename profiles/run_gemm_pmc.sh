#!/bin/bash
# PMC counters of the bf16 encoder GEMM (K = D shapes) and of the matrix-pipe scan kernel: bash profiles/run_gemm_pmc.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
SET2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"
run() {  # name, counter set, command...
  name=$1; shift; set_=$1; shift
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- "$@" > $O/$name.out 2>&1
  f=$(find /tmp/pmcx -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/gpu/gpu_pmc_summary.py $f > $O/$name.txt 2>&1
  grep -E "gemm|scan" $O/$name.txt
}
run pmc_gemm16_c3_set1 "$SET" python $R/tools/gpu/gpu_gemm16_one.py 3 8192 512 1120 30
run pmc_gemm16_c3_set2 "$SET2" python $R/tools/gpu/gpu_gemm16_one.py 3 8192 512 1120 30
run pmc_gemm16_c2_set1 "$SET" python $R/tools/gpu/gpu_gemm16_one.py 3 8192 512 320 30
run pmc_gemm16_sq4096_set1 "$SET" python $R/tools/gpu/gpu_gemm16_one.py 0 4096 4096 4096 10
VAMBHIP_SCAN_DBG=1 run pmc_scan_mfma_k32_set1 "$SET" python $R/tools/gpu/gpu_scan_one.py 2000000 32 32 20
VAMBHIP_SCAN_DBG=1 run pmc_scan_mfma_k32_set2 "$SET2" python $R/tools/gpu/gpu_scan_one.py 2000000 32 32 20
