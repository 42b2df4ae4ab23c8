#!/bin/bash
# Round 6, GPU call 34: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host-coherent memory: a runtime switch that
# shortens every kernel's start) -- the C2 training step and a C2 sweep with the variable 0 / 1 / unset
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y10; mkdir -p $O; cd $R
for v in unset 0 1 unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> $O/step_kernarg.txt
  timeout 300 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "" 2 2>&1 | grep SUMMARY >> $O/step_kernarg.txt
done
cat $O/step_kernarg.txt
for v in 0 1; do
  export HIP_FORCE_DEV_KERNARG=$v
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> $O/sweep_kernarg.txt
  VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1;X=1" >> $O/sweep_kernarg.txt 2>&1
done
grep "== HIP\|setting\|generator: total" $O/sweep_kernarg.txt | cut -c1-260
