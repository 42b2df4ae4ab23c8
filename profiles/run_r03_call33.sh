#!/bin/bash
# round 3, call 33: loss kernel forced to 8 waves per SIMD (50 VGPRs instead of 95): step time + per-kernel averages
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zf
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout 300 python tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 2>&1 | tail -1 | tee -a $O/epoch_time.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
    python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
grep -E "loss16|dz16|dadapt16|sum of kernel" $O/step_timeline.txt | tail -5 | cut -c1-150
