#!/bin/bash
# Round 4, final evidence on one MI355X (bash profiles/run_r04_final.sh; ~15 GPU-minutes):
#   1. the whole GPU test suite (incl. the 10 M x 64 sweep)
#   2. bf16 at scale: the product at C2 (2 M x 200, 300 epochs, same seeds) in fp32 and in bf16 --
#      final loss, clusters, ARI / purity / recovered genomes against the synthetic genomes
#   3. the bench line (C2 headline + c1 + c3_shape + cpu_baseline)
#   4. rocprofv3 --kernel-trace --stats of a short bench command (training only: the sweep's 0.3 M launches per job take the
#      profiler minutes to write) -- the encoder GEMM's average duration must agree with the bench probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for dt in fp32 bf16; do
  timeout 400 python tools/gpu/gpu_e2e_quality.py 2000000 200 300 8192 '[]' $dt 1 1 $O/e2e_quality_c2.jsonl > /dev/null 2> $O/e2e_$dt.err
done
python - <<'PY'
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04z", "e2e_quality_c2.jsonl")
for line in open(p):
    q = json.loads(line)
    print({k: q[k] for k in q if k != "loss_curve"})
PY
timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --epochs 20 --no-cluster --no-c3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_train20.csv && head -12 $f | cut -c1-160
rm -rf $O/prof
