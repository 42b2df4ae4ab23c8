#!/bin/bash
# Round 5, GPU call 12: the fp32 GEMM with four K-tiles in flight at the joint TaxVamb step's shapes; the joint trainer with it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
timeout 300 python tools/gpu/gpu_gemm_small.py $O/gemm_small.txt 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_vae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
for v in "VAMBHIP_VAE_GEMM_PREFETCH=4" "VAMBHIP_VAE_GEMM_PREFETCH=1"; do
  echo "== $v" >> $O/taxvamb_prefetch.txt
  env $v timeout 300 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 >> $O/taxvamb_prefetch.txt 2>&1
done
grep -v amdgpu.ids $O/taxvamb_prefetch.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tv -o bench -- python $R/tools/gpu/gpu_taxvamb_bench.py 50000 50 1000 > $O/taxvamb_profiled.txt 2>&1
f=$(find $O/prof_tv -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_taxvamb.csv
rm -rf $O/prof_tv
head -12 $O/kernel_stats_taxvamb.csv | cut -c1-200
