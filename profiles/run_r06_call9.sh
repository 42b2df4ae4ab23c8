#!/bin/bash
# Round 6, GPU call 9: step timeline under the new defaults; fork plan bit 16 (no fork at the top decoder layer)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
F="VAMBHIP_VAE_FORK_PLAN"
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "|$F=18|$F=19|$F=26|$F=3" 3 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "|$F=22|$F=18|$F=2" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 6 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1; sed -n 1,40p $O/step_timeline_C2.txt | cut -c1-140
rm -rf $O/prof
