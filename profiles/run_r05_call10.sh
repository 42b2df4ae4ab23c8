#!/bin/bash
# Round 5, GPU call 10: which variant of the bf16 step broke bit-identity in call 9 (tests/test_vae_gpu.py, scheduling variants)?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 500 python tools/gpu/gpu_variant_bits.py bf16 3 > $O/variant_bits.txt 2>&1; cat $O/variant_bits.txt | grep -v amdgpu.ids
VAMBHIP_VAEVAE_LANES=0 timeout 200 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 > $O/taxvamb_shared.txt 2>&1; cut -c1-400 $O/taxvamb_shared.txt | grep -v amdgpu.ids
