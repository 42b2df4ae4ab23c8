#!/bin/bash
# round 3, call 37: optimiser split (decoder-side half on the side stream): parity / determinism tests + C2 step time A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zj
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_semisup_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -k "not gemm" > $O/pytest_vae.log 2>&1; tail -2 $O/pytest_vae.log
for rep in 1 2; do
for v in 1 0; do
  VAMBHIP_VAE_OPT_SPLIT=$v timeout 300 python tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 2>&1 | tail -1 | sed "s/^/opt_split=$v: /" | tee -a $O/opt_split.txt
done; done
