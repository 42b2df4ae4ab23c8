#!/bin/bash
# tiles 3 (BK=32) and 4 (BK=64) on the C1 shapes: a_kc b_kc M N K splits
for t in 3 4; do for shp in "1 1 4096 512 512 1" "1 0 4096 512 512 1" "0 0 512 512 4096 4" "1 1 4096 512 4096 1"; do echo -n "tile $t $shp: "; python tools/gpu/gpu_gemm_probe.py $t $shp 5 | tail -1; done; done
