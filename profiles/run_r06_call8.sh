#!/bin/bash
# Round 6, GPU call 8: the pruned build: whole GPU suite + a short bench (1 timed job) for the launch count / kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 20 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_train20.csv && head -30 $f | cut -c1-130
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1; sed -n 1,45p $O/step_timeline_C2.txt | cut -c1-140
rm -rf $O/prof
