#!/bin/bash
# rocprofv3 kernel stats of the cluster kernels only, over a full C2 sweep on trained latents
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc2
timeout 900 rocprofv3 --kernel-trace --stats --kernel-include-regex "clu_" --output-format csv -d /tmp/pc2 -o t -- \
    python $R/tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PROFILE=1" > $O/c2_sweep_under_rocprof.txt 2>&1
f=$(find /tmp/pc2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_c2_cluster.csv
sed 's/(float const.*)",/",/; s/(int, unsigned.*)",/",/; s/(unsigned.*)",/",/' $O/kernel_stats_c2_cluster.csv | cut -c1-150
grep -E "setting|generator:" $O/c2_sweep_under_rocprof.txt | cut -c1-300
