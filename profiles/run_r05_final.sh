#!/bin/bash
# Round 5, final evidence on one MI355X (bash profiles/run_r05_final.sh; ~20 GPU-minutes):
#   1. __graft_entry__.smoke() and the whole GPU test suite on the round's last build
#   2. the bench line as the driver runs it (python bench.py: C2 headline + cpu_baseline + taxvamb + c1 + c3_shape legs)
#   3. rocprofv3 --kernel-trace --stats of a short bench command (training only: the sweep's 0.5 M launches per job take the
#      profiler minutes to write) + one step's kernel timeline -- the encoder GEMM's average duration must agree with the bench probe
#   4. PMC passes of the roofline kernel (encoder layer 0, K = D) at the C2 / C3 shapes: FETCH_SIZE and WRITE_SIZE in separate passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
SECONDS=0
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $SECONDS s" | tee $O/bench_wall.txt; tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 20 --no-cluster --no-c3 --no-taxvamb --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_train20.csv && head -14 $f | cut -c1-170
t=$(find $O/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline_C2.txt 2>&1
rm -rf $O/prof
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
  rm -f $O/pmc_fetch_$1.csv $O/pmc_write_$1.csv
done
