#!/bin/bash
# Round 6, GPU call 1: this round's box baseline of the round-5 build -- whole GPU suite, step time at C2 / C3 shape,
# per-kernel durations of the training leg, one C2 sweep with the generator's own profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_SINGLE_STREAM=1" 2 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "VAMBHIP_GEN_PREFILL=2" $O/sweep.json > $O/sweep.txt 2>&1; grep -v "passes with" $O/sweep.txt | grep -v amdgpu.ids | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train6.csv && head -24 $f | cut -c1-150
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1; head -60 $O/step_timeline.txt
rm -rf $O/prof
