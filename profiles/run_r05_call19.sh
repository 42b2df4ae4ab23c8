#!/bin/bash
# Round 5, GPU call 19: kernel profile of the joint TaxVamb step with the K-group GEMMs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tv -o bench -- python $R/tools/gpu/gpu_taxvamb_bench.py 50000 50 1000 > $O/taxvamb_profiled.txt 2>&1
f=$(find $O/prof_tv -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_taxvamb_kgroups.csv
rm -rf $O/prof_tv
head -30 $O/kernel_stats_taxvamb_kgroups.csv | cut -c1-200
