#!/bin/bash
# Round 6, GPU call 29: (1) counters of the row-major many-medoid scan (K6r) at 620 k rows x 32 medoids and of the one-medoid VALU pass
# at 100 k rows: SQ view, FETCH_SIZE, WRITE_SIZE in separate passes; (2) C2 sweeps with the two-step publish behind every pass
# (scan.publish_split = 2) against the default (passes with more than 8 medoids only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
pass() {  # name, counters, n, k
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/gpu/gpu_scan_one.py $3 32 $4 20 > $O/$1.out 2>&1
  f=$(find /tmp/pmcx -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/$1.csv
}
for cfg in "k6r 620000 32 clu_scan_mfma_rm_kernel" "valu1 100000 1 clu_scan_kernel"; do
  set -- $cfg
  pass sq_$1 "$SQ" $2 $3
  python $R/tools/gpu/gpu_pmc_summary.py $O/sq_$1.csv scan > $O/pmc_sq_scan_$1.txt 2>&1; cat $O/pmc_sq_scan_$1.txt | cut -c1-400
  pass fetch_$1 FETCH_SIZE $2 $3
  pass write_$1 WRITE_SIZE $2 $3
  python $R/tools/gpu/gpu_pmc_traffic.py $O/fetch_$1.csv $O/write_$1.csv $4 $O/pmc_traffic_scan_$1.json | cut -c1-400
  rm -f $O/sq_$1.csv $O/fetch_$1.csv $O/write_$1.csv
done
cd $R
S=VAMBHIP_SCAN_PUBLISH_SPLIT
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=1;$S=2;$S=1;$S=2" > $O/sweep_publish_split_all.txt 2>&1
grep "setting\|generator: total\|with  1 medoids\|with  4 medoids\|with  8 medoids" $O/sweep_publish_split_all.txt | cut -c1-260
