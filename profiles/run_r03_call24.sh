#!/bin/bash
# round 3, call 24: per-kernel times of one C2 job's sweep under the lazy-validation policy (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o sweep -- python $R/tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1" $O/sweep.json > $O/sweep.txt 2> $O/sweep.err
grep "vambhip\]" $O/sweep.err | cut -c1-900
cat $O/sweep.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python $R/tools/gpu/gpu_prof_summary.py $f 2>/dev/null | head -40 || head -30 $f
cp $f $O/kernel_stats_c2_job_lazy.csv
rm -rf $O/prof
