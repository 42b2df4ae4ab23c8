#!/bin/bash
# Round 6, GPU call 26: (1) the fp32 fault after the fix of the deep-prefetch GEMM's final wait (240 runs of 500 steps at the C2 shape;
# before: 5 faults in 244); (2) the row-major scan with the query block requested IN FRONT of the tiles: phase stamps old / new,
# C2 sweeps old / new on this box, the cluster / determinism / parallel tests on the new kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y2; mkdir -p $O; cd $R
bad=0
for i in $(seq 1 24); do
  timeout 200 python tools/gpu/gpu_fault_repro.py fp32 10 > $O/repro_tmp.log 2>&1 || { bad=$((bad+1)); echo "--- process $i" >> $O/repro_fixed.log; tail -3 $O/repro_tmp.log >> $O/repro_fixed.log; }
done
echo "fixed library: $bad of 24 processes (10 runs of 500 fp32 steps each) faulted; non-identical runs: $(grep -c 'identical=False' $O/repro_tmp.log)" | tee $O/repro_summary.txt
for lib in timing_old timing; do
  export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_$lib.so
  for shape in "170000 32 9" "170000 32 16" "620000 32 32" "2000000 32 32"; do
    echo "#### library $lib" >> $O/scan_timeline_k6r.txt
    timeout 200 python tools/gpu/gpu_scan_timeline.py $shape >> $O/scan_timeline_k6r.txt 2>&1
  done
done
unset VAMBHIP_LIB_PATH
grep -v "amdgpu.ids" $O/scan_timeline_k6r.txt | grep "####\|n=\|prologue\|row loop\|flush retired\|publish done\|wall" | cut -c1-200
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_determinism_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > $O/pytest_cluster.log 2>&1; tail -3 $O/pytest_cluster.log | cut -c1-300
for lib in oldpro new oldpro new; do
  if [ $lib = new ]; then unset VAMBHIP_LIB_PATH; else export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_$lib.so; fi
  echo "== library: $lib" >> $O/sweep_ab.txt
  VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1;X=1" >> $O/sweep_ab.txt 2>&1
done
unset VAMBHIP_LIB_PATH
grep "== library\|setting\|generator: total\|with 32 medoids\|with 16 medoids\|with  9 medoids" $O/sweep_ab.txt | cut -c1-300
