#!/bin/bash
# round 3, call 21: label models (VAEConcat / VAELabels) parity + regression of the base VAE golden tests
mkdir -p gpurun_out/r03u
timeout 1200 python -m pytest tests/test_semisup_gpu.py -m gpu -q -x > gpurun_out/r03u/pytest_semisup.log 2>&1
tail -25 gpurun_out/r03u/pytest_semisup.log
timeout 1200 python -m pytest tests/test_vae_gpu.py tests/test_dp_gpu.py tests/test_parallel_gpu.py -m gpu -q -x -k "not gemm" > gpurun_out/r03u/pytest_vae.log 2>&1
tail -5 gpurun_out/r03u/pytest_vae.log
