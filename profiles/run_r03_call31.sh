#!/bin/bash
# round 3, call 31: bias-gradient column sums inside the weight-gradient GEMM (vae.dz_colsum=0, new default) vs in the dz kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zd
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_semisup_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "not gemm" > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log
for rep in 1 2; do
for v in 0 1; do
  VAMBHIP_VAE_DZ_COLSUM=$v timeout 300 python tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 2>&1 | tail -1 | sed "s/^/dz_colsum=$v: /" | tee -a $O/dz_colsum.txt
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
    python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
tail -22 $O/step_timeline.txt | cut -c1-150
