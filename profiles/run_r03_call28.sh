#!/bin/bash
# round 3, call 28: C2 step time + per-kernel timeline after the loss-reduction fix (no scratch) and the fork at the loss kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03za
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout 300 python tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 2>&1 | tail -1 | tee -a $O/epoch_time.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
    python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_e3.csv
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
head -36 $O/step_timeline.txt | cut -c1-110; tail -24 $O/step_timeline.txt | cut -c1-150
