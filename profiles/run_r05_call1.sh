#!/bin/bash
# Round 5, GPU call 1 (bash profiles/run_r05_call1.sh): validate the first batch of step changes and measure them on ONE box:
#   fused latent-wide kernels (gemm_skinny16.hpp), epilogue operands requested in the GEMM prologue, optimiser table lookup,
#   scalar tail of the optimiser on the last workgroup.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
# 1. the VAE tests first (bit-identity of the fused paths against the split launches), then the rest of the GPU suite
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_vae.log 2>&1; tail -4 $O/pytest_vae.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --ignore=tests/test_vae_gpu.py > $O/pytest_rest.log 2>&1; tail -4 $O/pytest_rest.log
# 2. step time at C2: the round-4 library (built from HEAD~ sources into .r4base/) and this build with its toggles
if [ -f .r4base/vamb_amd/libvambhip.so ]; then
  timeout 300 python .r4base/tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 > $O/step_r4_c2.txt 2>&1; tail -1 $O/step_r4_c2.txt
fi
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_FUSED_SKINNY=0|VAMBHIP_VAE_FUSED_FINALIZE=0|VAMBHIP_VAE_FUSED_SKINNY=0;VAMBHIP_VAE_FUSED_FINALIZE=0|VAMBHIP_SINGLE_STREAM=1" 2 > $O/step_ab_c2.txt 2>&1; grep SUMMARY $O/step_ab_c2.txt
# 2b. where the elementwise BatchNorm-backward kernel's time goes (timing experiments, wrong results): no atomics / no stores / no statistics loads
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 200 8192 8 bf16 "|VAMBHIP_VAE_DZ_DBG=1|VAMBHIP_VAE_DZ_DBG=2|VAMBHIP_VAE_DZ_DBG=4|VAMBHIP_VAE_DZ_DBG=7" 1 > $O/step_dz_dbg.txt 2>&1; grep SUMMARY $O/step_dz_dbg.txt
# 2c. in-kernel phases of the encoder GEMM, before / after the prologue requests its epilogue operands
if [ -f .r4base/vamb_amd/libvambhip.so ]; then timeout 200 python .r4base/tools/gpu/gpu_gemm16_timeline.py $O/gemm16_timeline_r4.txt > /dev/null 2>&1; fi
timeout 200 python tools/gpu/gpu_gemm16_timeline.py $O/gemm16_timeline.txt > /dev/null 2>&1; grep "epi 3 variant 21" $O/gemm16_timeline_r4.txt $O/gemm16_timeline.txt
# 3. the north-star shape (2 M x 1000)
if [ -f .r4base/vamb_amd/libvambhip.so ]; then
  timeout 400 python .r4base/tools/gpu/gpu_epoch_time.py 2000000 1000 8192 6 bf16 > $O/step_r4_c3.txt 2>&1; tail -1 $O/step_r4_c3.txt
fi
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|VAMBHIP_VAE_FUSED_SKINNY=0;VAMBHIP_VAE_FUSED_FINALIZE=0" 2 > $O/step_ab_c3.txt 2>&1; grep SUMMARY $O/step_ab_c3.txt
# 4. per-kernel durations of the training leg (this build, defaults)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train6.csv && head -30 $f | cut -c1-150
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1; head -50 $O/step_timeline.txt
rm -rf $O/prof
