#!/bin/bash
# Round 3, GPU call 1: hardware probes (ds_read_b64_tr_b16 lane mapping, L2 -> CU operand feed by path), bf16 error
# diagnostics, the GPU suite on the advisor fixes, and the end-to-end quality runs next to the reference's.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 60 ./tests/micro/build/tr16_probe > $O/tr16_probe.txt 2>&1; tail -5 $O/tr16_probe.txt
for K in 320 512 1120; do timeout 120 ./tests/micro/build/loadpath $K > $O/loadpath_k$K.json 2>&1; done
cat $O/loadpath_k512.json
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -4 $O/pytest_gpu.log
VAMBHIP_PRECISION=bf16 timeout 300 python tests/diagnostics/gpu_large_batch_errors.py 8192 > $O/bf16_errors_b8192.txt 2>&1; head -24 $O/bf16_errors_b8192.txt
timeout 300 python tests/diagnostics/gpu_bf16_errors.py bf16 > $O/bf16_errors_small.txt 2>&1
Q=$O/e2e_quality.jsonl
for dt in fp32 bf16; do for seed in 0 1; do
  timeout 300 python tools/gpu/gpu_e2e_quality.py 20000 50 30 256 '[3,8,15,22]' $dt $seed 1 $Q | cut -c1-400
done; done
for dt in bf16 fp32; do
  timeout 600 python tools/gpu/gpu_e2e_quality.py 100000 200 300 4096 '[]' $dt 0 2 $Q | cut -c1-400
done
timeout 900 python tools/gpu/gpu_e2e_quality.py 2000000 200 300 8192 '[]' bf16 0 1 $Q | cut -c1-600
