#!/bin/bash
# Round 5, GPU call 2: tests of the joint-trainer binding fix; per-kernel profile of the step (C2 and the C3 shape);
# where the 32-medoid scan pass spends its time; a C2 sweep profile as the baseline of the sweep work.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_vae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
timeout 300 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|VAMBHIP_VAE_FUSED_SKINNY=0" 2 > $O/step_ab_c2.txt 2>&1; grep SUMMARY $O/step_ab_c2.txt
timeout 300 python tools/gpu/gpu_scan_dbg.py 620000 32 $O/scan_dbg_k32.txt > /dev/null 2>&1; cat $O/scan_dbg_k32.txt
timeout 300 python tools/gpu/gpu_scan_dbg.py 170000 16 $O/scan_dbg_k16_small.txt > /dev/null 2>&1; cat $O/scan_dbg_k16_small.txt
VAMBHIP_GEN_PROFILE=1 timeout 600 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "X=1" $O/sweep_base.json > $O/sweep_base.txt 2>&1; grep -v "passes with" $O/sweep_base.txt | tail -8
cd /tmp && export TMPDIR=/tmp
for cfg in "C2 200" "C3 1000"; do set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o bench -- python $R/tools/gpu/gpu_epoch_time.py 2000000 $2 8192 3 bf16 > $O/epoch_profiled_$1.txt 2>&1
  f=$(find $O/prof_$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  t=$(find $O/prof_$1 -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_$1.txt 2>&1
  rm -rf $O/prof_$1
done
cat $O/step_timeline_C2.txt | head -70
