#!/bin/bash
# production tile (3 = 64x64) on the C1 shapes: a_kc b_kc M N K splits
for shp in "1 1 4096 512 512 1" "1 1 4096 512 160 1" "1 1 4096 160 512 1" "1 0 4096 512 512 1" "0 0 512 512 4096 4" "0 0 512 160 4096 8" "1 1 4096 512 4096 1"; do echo -n "tile 3 $shp: "; python tests/gpu_gemm_probe.py 3 $shp 5 | tail -1; done
