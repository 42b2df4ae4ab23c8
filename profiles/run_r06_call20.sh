#!/bin/bash
# Round 6, GPU call 20: trimmed training epilogue, second build (the no-dropout path keeps the general epilogue's arithmetic): VAE-side GPU tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_semisup_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py tests/test_cli_gpu.py -m gpu -q > $O/pytest_vae.log 2>&1; tail -3 $O/pytest_vae.log | cut -c1-300
