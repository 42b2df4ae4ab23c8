#!/bin/bash
# Round 6, GPU call 30: soak of the last build -- the training loops (fp32 and bf16, 500 steps at the C2 shape, 120 + 80 runs), the
# TaxVamb / semi-supervised tests five times (the K-group and deep-prefetch fp32 tiles at batch 256), the cluster determinism test
# five times; every process outside pytest's fd capture or with -s so that a runtime message would be seen
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y6; mkdir -p $O; cd $R
bad=0
for i in $(seq 1 12); do timeout 200 python tools/gpu/gpu_fault_repro.py fp32 10 > $O/tmp.log 2>&1 || { bad=$((bad+1)); tail -3 $O/tmp.log >> $O/soak_faults.log; }; done
echo "fp32: $bad of 12 processes x 10 runs of 500 steps faulted" | tee -a $O/soak_summary.txt
bad=0
for i in $(seq 1 8); do timeout 200 python tools/gpu/gpu_fault_repro.py bf16 10 > $O/tmp.log 2>&1 || { bad=$((bad+1)); tail -3 $O/tmp.log >> $O/soak_faults.log; }; done
echo "bf16: $bad of 8 processes x 10 runs of 500 steps faulted; non-identical: $(grep -c 'identical=False' $O/tmp.log)" | tee -a $O/soak_summary.txt
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_determinism_gpu.py -m gpu -q -s -x > $O/tmp_pytest.log 2>&1
  echo "pass $i: $(grep -E 'passed|failed|error' $O/tmp_pytest.log | tail -1) $(grep -c -i 'memory access fault' $O/tmp_pytest.log) fault message(s)" | tee -a $O/soak_summary.txt
done
