#!/bin/bash
# Round 5, GPU call 21: K groups with uneven slabs / deep prefetch with remainder iterations: tests, shapes, the joint step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_vaevae_gpu.py tests/test_semisup_gpu.py -m gpu -q --maxfail=8 > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
timeout 300 python tools/gpu/gpu_gemm_small.py $O/gemm_small.txt 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-330
for v in "VAMBHIP_VAE_GEMM_KGROUPS=4" "VAMBHIP_VAE_GEMM_KGROUPS=1"; do
  echo "== $v" >> $O/taxvamb_kgroups_uneven.txt
  env $v timeout 300 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 >> $O/taxvamb_kgroups_uneven.txt 2>&1
done
grep -v amdgpu.ids $O/taxvamb_kgroups_uneven.txt | cut -c1-330
timeout 300 python tools/gpu/gpu_epoch_time.py 200000 50 4096 20 fp32 2>&1 | tail -3
