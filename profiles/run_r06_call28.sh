#!/bin/bash
# Round 6, GPU call 28: (1) the candidate walk asks for its round's statistics ONCE (until now in front of every candidate: 25 + 24 +
# ... hash lookups per failed round, on the host's critical path): C2 sweeps, per-step variant library against the new one;
# (2) passes with more than 8 medoids publish their four summary words first and the histograms behind a second flag
# (scan.publish_split): phase stamps and C2 sweeps with the option on / off; cluster / determinism / parallel / e2e tests on the build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_determinism_gpu.py tests/test_parallel_gpu.py tests/test_e2e_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/pytest_cluster.log 2>&1; tail -2 $O/pytest_cluster.log | cut -c1-200
export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_timing.so
for sp in 0 1; do
  for shape in "170000 32 16" "620000 32 32"; do
    echo "#### scan.publish_split = $sp" >> $O/publish_split.txt
    VAMBHIP_SCAN_PUBLISH_SPLIT=$sp timeout 200 python tools/gpu/gpu_scan_timeline.py $shape >> $O/publish_split.txt 2>&1
  done
done
unset VAMBHIP_LIB_PATH
grep "####\|n=\|flush retired\|publish\|wall\|last such" $O/publish_split.txt | cut -c1-150
for lib in everystep new everystep new; do
  if [ $lib = new ]; then unset VAMBHIP_LIB_PATH; else export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_$lib.so; fi
  echo "== library: $lib" >> $O/sweep_ab.txt
  VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1;X=1" >> $O/sweep_ab.txt 2>&1
done
unset VAMBHIP_LIB_PATH
grep "== library\|setting\|generator: total\|host time" $O/sweep_ab.txt | cut -c1-420
S=VAMBHIP_SCAN_PUBLISH_SPLIT
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=1;$S=0;$S=1;$S=0" > $O/sweep_publish_split.txt 2>&1
grep "setting\|generator: total\|with 32 medoids\|with 16 medoids" $O/sweep_publish_split.txt | cut -c1-260
