#!/bin/bash
# Round 3, GPU call 9: full GPU suite on the round's changes (new e2e / C4 / 2-process tests included) + bench.py end to end
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -25 $O/pytest_gpu.log
(timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','epoch_ms','us_per_train_step','train_s','encode_ms','cluster_ms','clusters_per_step','final_loss'):
    print(k, d.get(k))
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','avg_launch_ms','traffic_source')})
print('cluster_scan', d['cluster_scan'])
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:600])
print('c3_shape', json.dumps(d.get('c3_shape'))[:2500])
PY
tail -3 $O/bench.err
