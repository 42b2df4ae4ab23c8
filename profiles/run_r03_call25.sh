#!/bin/bash
# round 3, call 29: side stream forked at the loss kernel vs after the first decoder dz kernel (C2 step time, 3 repeats each)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zb
mkdir -p $O
cd $R
for rep in 1 2; do
for v in 1 0; do
  VAMBHIP_VAE_FORK_AT_LOSS=$v timeout 300 python tools/gpu/gpu_epoch_time.py 2000000 200 8192 12 bf16 2>&1 | tail -1 | sed "s/^/fork_at_loss=$v: /" | tee -a $O/fork_at_loss.txt
done; done
