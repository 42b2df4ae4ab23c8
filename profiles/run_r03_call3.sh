#!/bin/bash
# Round 3, GPU call 3: in-kernel timeline of the bf16 GEMM; step timeline (rocprofv3 kernel trace) of the new dataflow.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 300 python tools/gpu/gpu_gemm16_timeline.py $O/gemm16_timeline.txt
cd /tmp && export TMPDIR=/tmp
for tag in p2 p0; do
  rm -rf /tmp/prof
  opt=""; [ $tag = p0 ] && opt="VAMBHIP_VAE_GEMM_PIPELINE=0"
  env $opt timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
      python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof_$tag.json 2> $O/prof_$tag.err
  f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_e3_$tag.csv
  t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_$tag.txt 2>&1
  tail -48 $O/step_timeline_$tag.txt | cut -c1-150
done
