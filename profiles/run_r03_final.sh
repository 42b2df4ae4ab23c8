#!/bin/bash
# Round 3: final evidence -- GPU suite, bench.py as the driver runs it (fewer steps), rocprofv3 kernel stats of a short run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03z}
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); grep -v "INFO " $O/pytest_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
(timeout 1700 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err); tail -2 $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','epoch_ms','us_per_train_step','train_s','encode_ms','cluster_ms','clusters_per_step','final_loss'):
    print(k, d.get(k))
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','avg_launch_ms','traffic')}, d['roofline'].get('hbm_side'))
print('cluster_scan', {k:v for k,v in d['cluster_scan'].items() if k not in ('bytes','measured_in')})
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
c=d['c3_shape']; print('c3', {k:c.get(k) for k in ('value','job_s','train_s','cluster_s','clusters','epoch_ms','us_per_step')}, c['roofline_encoder_gemm']['frac'], c['cluster_scan']['frac'], c['cluster_scan']['frac_over_sweep_wall_time'])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
    python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_e3.csv
t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
tail -24 $O/step_timeline.txt | cut -c1-150
