#!/bin/bash
# Round 6, final evidence of the LAST build on one MI355X (bash profiles/run_r06_final6.sh; ~25 GPU-minutes):
#   1. __graft_entry__.smoke() and the whole GPU test suite on the round's last build
#   2. the bench line as the driver runs it (python bench.py: C2 headline + cpu_baseline + taxvamb + c1 + c3_shape legs)
#   3. rocprofv3 --kernel-trace --stats of a short bench command (training only) + one step's kernel timeline: the K = D encoder GEMM
#      now has its own kernel name (gemm_bf16_kernel<..., 3, 2, 1>), so its average duration in the trace IS the roofline kernel's
#   4. PMC passes of that kernel shape (encoder layer 0, K = D) at C2 / C3: FETCH_SIZE and WRITE_SIZE in separate passes, and the SQ view
#      (matrix-pipe busy, wait states) VERDICT r5 item 3 asks for
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f6; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
SECONDS=0
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $SECONDS s" | tee $O/bench_wall.txt; tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 20 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_train20.csv && head -14 $f | cut -c1-170
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1
rm -rf $O/prof
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
pass() {  # name, counters, K
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/gpu/gpu_gemm16_one.py 3 8192 512 $3 30 > $O/$1.out 2>&1
  f=$(find /tmp/pmcx -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/$1.csv
}
for cfg in "c2 320" "c3 1120"; do
  set -- $cfg
  pass pmc_fetch_$1 FETCH_SIZE $2
  pass pmc_write_$1 WRITE_SIZE $2
  python $R/tools/gpu/gpu_pmc_traffic.py $O/pmc_fetch_$1.csv $O/pmc_write_$1.csv "gemm_bf16_kernel<128, 128, 2, 4, 3, 2" $O/pmc_roofline_$1.json
  cat $O/pmc_roofline_$1.json | cut -c1-400
  pass pmc_sq_gemm16_$1 "$SET" $2
  python $R/tools/gpu/gpu_pmc_summary.py $O/pmc_sq_gemm16_$1.csv > $O/pmc_sq_gemm16_$1.txt 2>&1; grep gemm $O/pmc_sq_gemm16_$1.txt | cut -c1-400
  rm -f $O/pmc_fetch_$1.csv $O/pmc_write_$1.csv $O/pmc_sq_gemm16_$1.csv
done
# a second pass of the whole GPU suite on the same box (the round found a fault that showed in ~2 % of the runs of one test)
cd $R; timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu_second_pass.log 2>&1; tail -2 $O/pytest_gpu_second_pass.log | cut -c1-200
