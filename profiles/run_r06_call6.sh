#!/bin/bash
# Round 6, GPU call 6: fork plans around the paired weight-gradient launch (plan 2 = small kernels FIRST on the side stream), with the
# optimiser's tail behind two-level tickets, at C2 and the C3 shape; timeline of the best
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
P="VAMBHIP_VAE_DW_PAIR"; F="VAMBHIP_VAE_FORK_PLAN"; Z="VAMBHIP_VAE_FUSED_FINALIZE"; L="VAMBHIP_VAE_FORK_AT_LOSS"
timeout 900 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$F=2|$F=2;$Z=1|$F=2;$P=0|$F=0|$F=3|$F=2;$L=0|$F=0;$Z=1|$F=3;$Z=1" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$F=2|$F=2;$Z=1|$F=0;$Z=1|$F=3;$Z=1" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
cd /tmp && export TMPDIR=/tmp
VAMBHIP_VAE_FORK_PLAN=2 VAMBHIP_VAE_FUSED_FINALIZE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train6.csv
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1; sed -n 1,45p $O/step_timeline.txt | cut -c1-160
rm -rf $O/prof
