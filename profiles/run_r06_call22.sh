#!/bin/bash
# Round 6, GPU call 22: vae_dz16_kernel without per-element control flow (MASKED template, pair packing): VAE tests, step A/B against the
# previous build (VAMBHIP_LIB_PATH) at C2 and the C3 shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_semisup_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py -m gpu -q > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log | cut -c1-300
for lib in prev new prev new; do
  if [ $lib = prev ]; then export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_prev.so; else unset VAMBHIP_LIB_PATH; fi
  echo "== library: $lib" >> $O/step_c2.txt; echo "== library: $lib" >> $O/step_c3.txt
  timeout 300 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "" 2 2>&1 | grep SUMMARY >> $O/step_c2.txt
  timeout 300 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "" 2 2>&1 | grep SUMMARY >> $O/step_c3.txt
done
cat $O/step_c2.txt $O/step_c3.txt
