#!/bin/bash
# Round 5, GPU call 17: the fp32 GEMM with four wavefront groups over K (tile 7) at the joint TaxVamb step's shapes; the joint step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R
timeout 300 python tools/gpu/gpu_gemm_small.py $O/gemm_small.txt 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 900 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_vae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
for v in "VAMBHIP_VAE_GEMM_KGROUPS=4" "VAMBHIP_VAE_GEMM_KGROUPS=1"; do
  echo "== $v" >> $O/taxvamb_kgroups.txt
  env $v timeout 300 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 >> $O/taxvamb_kgroups.txt 2>&1
done
grep -v amdgpu.ids $O/taxvamb_kgroups.txt | cut -c1-420
