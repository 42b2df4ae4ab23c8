#!/bin/bash
# Round 6, GPU call 23: (1) the final-evidence run of call "final3" ABORTED inside tests/test_determinism_gpu.py::
# test_500_training_steps_at_the_c2_shape_are_bit_identical[fp32] (pytest's fd capture swallowed the runtime's message): the same
# training loop outside pytest, many times, default and with one feature off at a time, to get the message and a frequency;
# (2) where the time of a scan pass goes (timing build: in-kernel stamps), four shapes of a C2 sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06v; mkdir -p $O; cd $R
run_cfg() {  # name, processes, dtype, env...
  name=$1; procs=$2; dt=$3; shift 3
  ok=0; bad=0
  for i in $(seq 1 $procs); do
    env "$@" timeout 200 python tools/gpu/gpu_fault_repro.py $dt 8 >> $O/repro_$name.log 2>&1
    rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "process $i rc=$rc" >> $O/repro_$name.log; fi
  done
  echo "$name ($dt; $*): $ok processes of 8 runs ok, $bad failed; non-identical runs: $(grep -c 'identical=False' $O/repro_$name.log)" | tee -a $O/repro_summary.txt
  grep -i "fault\|error" $O/repro_$name.log | sort | uniq -c | head -5 | tee -a $O/repro_summary.txt
}
run_cfg default_fp32 12 fp32 X=1
run_cfg default_bf16 4 bf16 X=1
run_cfg noprefetchbatch_fp32 5 fp32 VAMBHIP_VAE_PREFETCH_BATCH=0
run_cfg plaintiles_fp32 5 fp32 VAMBHIP_VAE_GEMM_PREFETCH=1 VAMBHIP_VAE_GEMM_KGROUPS=1
run_cfg singlestream_fp32 5 fp32 VAMBHIP_SINGLE_STREAM=1
export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_timing.so
for shape in "100000 32 1" "150000 32 4" "170000 32 16" "620000 32 32" "620000 32 8"; do
  timeout 200 python tools/gpu/gpu_scan_timeline.py $shape >> $O/scan_timeline.txt 2>&1
done
cat $O/scan_timeline.txt
