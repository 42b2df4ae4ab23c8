#!/bin/bash
# Round 5, GPU call 14: the joint TaxVamb step on TWO lanes (four streams) against the shared pair; step timeline with the fixed tool
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_vaevae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_vaevae.log 2>&1; tail -2 $O/pytest_vaevae.log
timeout 300 python -m pytest tests/test_cluster_gpu.py -m gpu -q -k "scan_accumulators_bit_exact" > $O/pytest_scan_shapes.log 2>&1; tail -2 $O/pytest_scan_shapes.log
for v in "VAMBHIP_VAEVAE_LANES=2" "VAMBHIP_VAEVAE_LANES=0" "VAMBHIP_VAEVAE_LANES=2"; do
  echo "== $v" >> $O/taxvamb_two_lanes.txt
  env $v timeout 300 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 >> $O/taxvamb_two_lanes.txt 2>&1
done
grep -v amdgpu.ids $O/taxvamb_two_lanes.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -o bench -- python $R/tools/gpu/gpu_epoch_time.py 2000000 200 8192 3 bf16 > $O/epoch_profiled_C2.txt 2>&1
t=$(find $O/prof_C2 -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1
rm -rf $O/prof_C2
head -48 $O/step_timeline_C2.txt
