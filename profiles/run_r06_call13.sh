#!/bin/bash
# Round 6, GPU call 13: BatchNorm fold inside the consuming GEMM (gemm_bf16.hpp STG 4) + the two-workgroups-per-CU tile of the K = D launch:
# parity tests, step A/B at C2 and the C3 shape, step timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_vae_gpu.py -m gpu -q -x > $O/pytest_vae.log 2>&1; tail -5 $O/pytest_vae.log | cut -c1-300
F="VAMBHIP_VAE_FOLD_IN_GEMM"
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$F=0" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$F=0" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1; sed -n 1,45p $O/step_timeline_C2.txt | cut -c1-140
rm -rf $O/prof
