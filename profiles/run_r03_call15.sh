#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_e2e_gpu.py tests/test_dp_gpu.py tests/test_prep_gpu.py -m gpu -q -x > $O/pytest_vae.log 2>&1; echo "rc=$?" >> $O/pytest_vae.log); grep -v "INFO " $O/pytest_vae.log | tail -6
for args in "bf16 3 epochs 9 0.2"; do timeout 200 python tools/gpu/gpu_determinism.py $args 2>&1 | grep -v amdgpu; done
for opt in "A=default" "VAMBHIP_VAE_PREFETCH_BATCH=0"; do
  for rep in 1 2; do
    echo -n "[$opt] " | tee -a $O/step_time_ab.txt
    env $opt timeout 300 python tools/gpu/gpu_epoch_time.py 400000 200 8192 40 bf16 2>/dev/null | tee -a $O/step_time_ab.txt
  done
done
