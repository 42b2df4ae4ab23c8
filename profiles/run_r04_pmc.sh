#!/bin/bash
# Round 4 (re-take on the final build; the kernel is unchanged since round 3): PMC counters of the production K = D encoder GEMM (lean epilogue, three-buffer K loop) at the C2 / C3 shapes:
# HBM-side traffic (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections in tools/gpu/gpu_pmc_traffic.py) and the
# SQ view (matrix-pipe busy, wait states).  bash profiles/run_r04_pmc.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
pass() {  # name, counters, K
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/gpu/gpu_gemm16_one.py 3 8192 512 $3 30 > $O/$1.out 2>&1
  f=$(find /tmp/pmcx -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/$1.csv
}
for cfg in "c2 320" "c3 1120"; do
  set -- $cfg
  pass pmc_fetch_$1 FETCH_SIZE $2
  pass pmc_write_$1 WRITE_SIZE $2
  python $R/tools/gpu/gpu_pmc_traffic.py $O/pmc_fetch_$1.csv $O/pmc_write_$1.csv "gemm_bf16_kernel<128, 128, 2, 4, 3, 2" $O/pmc_roofline_$1.json
  pass pmc_sq_$1 "$SET" $2
  python $R/tools/gpu/gpu_pmc_summary.py $O/pmc_sq_$1.csv > $O/pmc_sq_$1.txt 2>&1; grep gemm $O/pmc_sq_$1.txt | cut -c1-400
done
