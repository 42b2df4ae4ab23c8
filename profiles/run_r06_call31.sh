#!/bin/bash
# Round 6, GPU call 31: the candidate walk tests "already tried" with a byte per row instead of a list search per pool row: C2 sweeps,
# previous library against the new one, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y7; mkdir -p $O; cd $R
for lib in prev new prev new; do
  if [ $lib = new ]; then unset VAMBHIP_LIB_PATH; else export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_$lib.so; fi
  echo "== library: $lib" >> $O/sweep_ab.txt
  VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1;X=1" >> $O/sweep_ab.txt 2>&1
done
unset VAMBHIP_LIB_PATH
grep "== library\|setting\|generator: total\|host time" $O/sweep_ab.txt | cut -c1-420
timeout 600 python -m pytest tests/test_cluster_gpu.py tests/test_determinism_gpu.py tests/test_parallel_gpu.py -m gpu -q -x > $O/pytest_cluster.log 2>&1; grep -E "passed|failed" $O/pytest_cluster.log | tail -1
