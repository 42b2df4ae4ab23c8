#!/bin/bash
# Round 3, GPU call 2: the row-major weight-gradient GEMM (transposing LDS reads) and the three-buffer / interleaved-DMA K loop:
# numerics, variant timings, the VAE suite on the new dataflow, step time A/B through the library options.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -x -q -k "gemm16" > $O/pytest_gemm16.log 2>&1; echo "rc=$?" >> $O/pytest_gemm16.log); tail -5 $O/pytest_gemm16.log
timeout 600 python tools/gpu/gpu_gemm16_variants.py $O/gemm16_variants.json > $O/gemm16_variants.txt 2>&1; cat $O/gemm16_variants.txt
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -5 $O/pytest_gpu.log
for opt in "" "VAMBHIP_VAE_GEMM_PIPELINE=0" "VAMBHIP_VAE_DW_ROW_MAJOR=0" "VAMBHIP_VAE_GEMM_PIPELINE=0 VAMBHIP_VAE_DW_ROW_MAJOR=0"; do
  echo "== options: $opt"
  env $opt timeout 600 python bench.py --epochs 10 --steps 1 --warmup 1 --no-cpu-baseline --no-c3 --no-cluster 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')}, d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline',{}).get('frac'), {k:v for k,v in d.get('config',{}).items() if 'epoch' in k or 'step' in k})"
done
