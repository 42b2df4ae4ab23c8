#!/bin/bash
# Round 6, GPU call 4: paired launch of the last two weight gradients (vae.dw_pair) + third stream for the two one-workgroup kernels
# (vae.aux_stream): VAE / CLI / determinism tests, the start-up self-test, step A/B at C2 and the C3 shape, step timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_cli_gpu.py tests/test_determinism_gpu.py -m gpu -q --maxfail=8 > $O/pytest_vae.log 2>&1; tail -12 $O/pytest_vae.log | cut -c1-300
python -c "
from vamb_amd import _lib
_lib.require_gpu(); print('selftest fallbacks mask:', _lib.selftest_fallbacks)" 2>&1 | tail -3
timeout 500 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_DW_PAIR=0|VAMBHIP_VAE_AUX_STREAM=0|VAMBHIP_VAE_DW_PAIR=0;VAMBHIP_VAE_AUX_STREAM=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 500 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|VAMBHIP_VAE_DW_PAIR=0;VAMBHIP_VAE_AUX_STREAM=0" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train6.csv
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1; sed -n 1,70p $O/step_timeline.txt | cut -c1-160
rm -rf $O/prof
