#!/bin/bash
# Round 3, GPU call 4: lean epilogue -- numerics (GEMM + VAE suites), in-kernel timeline, step timeline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_dp_gpu.py -m gpu -x -q > $O/pytest_vae.log 2>&1; echo "rc=$?" >> $O/pytest_vae.log); tail -15 $O/pytest_vae.log
timeout 300 python tools/gpu/gpu_gemm16_timeline.py $O/gemm16_timeline.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
    python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_e3.csv
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
tail -48 $O/step_timeline.txt | cut -c1-150
cd $R
timeout 600 python bench.py --epochs 20 --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --no-cluster 2>/dev/null | tail -1 | cut -c1-1500
