#!/bin/bash
# Round 6, GPU call 2: dZ formed on the operand path of the input-gradient GEMM (vae.fused_dz): bit-identity tests, the CLI replay
# test, step A/B at C2 / the C3 shape, per-kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -q --maxfail=8 -k "fused_dz or scheduling_variants" > $O/pytest_fused_dz.log 2>&1; tail -5 $O/pytest_fused_dz.log
timeout 300 python -m pytest tests/test_cli_gpu.py -m gpu -q > $O/pytest_cli.log 2>&1; tail -15 $O/pytest_cli.log
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_FUSED_DZ=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|VAMBHIP_VAE_FUSED_DZ=0" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train6.csv && head -12 $f | cut -c1-150
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1; sed -n 1,50p $O/step_timeline.txt
rm -rf $O/prof
