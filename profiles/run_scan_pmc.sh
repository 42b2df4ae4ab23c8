#!/bin/bash
# PMC counters of the scan kernel at one shape: bash profiles/run_scan_pmc.sh TAG n L k
TAG=$1; N=$2; L=$3; K=$4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  VAMBHIP_SCAN_DBG=${SCAN_DBG:-0} timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -o pmc -- python $R/tools/gpu/gpu_scan_one.py $N $L $K 20 > $O/pmc$i.out 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/pmc_${K}_set$i.csv && python $R/tools/gpu/gpu_pmc_summary.py $f scan_kernel > $O/pmc_${K}_set$i.txt 2>&1
  cat $O/pmc_${K}_set$i.txt; tail -2 $O/pmc$i.out
done
