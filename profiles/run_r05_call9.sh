#!/bin/bash
# Round 5, GPU call 9: the joint TaxVamb trainer with one stream pair per pass (vaevae.lanes) against the shared pair; the bf16
# loss kernel with DPP wave reductions.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_vaevae_gpu.py tests/test_semisup_gpu.py tests/test_vae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
VAMBHIP_VAEVAE_LANES=0 timeout 300 python -m pytest tests/test_vaevae_gpu.py -m gpu -q --maxfail=8 > $O/pytest_vaevae_shared.log 2>&1; tail -2 $O/pytest_vaevae_shared.log
for v in "VAMBHIP_VAEVAE_LANES=1" "VAMBHIP_VAEVAE_LANES=0" "VAMBHIP_VAEVAE_LANES=1 GPU_MAX_HW_QUEUES=8"; do
  echo "== $v" >> $O/taxvamb_lanes.txt
  env $v timeout 300 python tools/gpu/gpu_taxvamb_bench.py 200000 50 1000 >> $O/taxvamb_lanes.txt 2>&1
done
cat $O/taxvamb_lanes.txt | cut -c1-600
timeout 300 python tools/gpu/gpu_step_ab.py 2000000 200 8192 12 bf16 "" 2 > $O/step_c2.txt 2>&1; grep SUMMARY $O/step_c2.txt
timeout 300 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 6 bf16 "" 2 > $O/step_c3.txt 2>&1; grep SUMMARY $O/step_c3.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tv -o bench -- python $R/tools/gpu/gpu_taxvamb_bench.py 50000 50 1000 > $O/taxvamb_profiled.txt 2>&1
f=$(find $O/prof_tv -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_taxvamb.csv
t=$(find $O/prof_tv -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python - "$t" > $O/taxvamb_trace_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
q = collections.Counter(r.get("Queue_Id", "?") for r in rows)
print("kernels", len(rows), "span ms", (t1 - t0) / 1e6, "sum of kernel time ms", busy / 1e6, "queues", dict(q))
# concurrency histogram over the last third of the trace (steady training at batch 4096 / 256 mix)
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = ev[0][0]
for ts, d in ev:
    hist[cur] += ts - last; last = ts; cur += d
tot = sum(hist.values())
print("time share by number of kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
PY
cat $O/taxvamb_trace_summary.txt
for cfg in "C2 200"; do set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o bench -- python $R/tools/gpu/gpu_epoch_time.py 2000000 $2 8192 3 bf16 > $O/epoch_profiled_$1.txt 2>&1
  f=$(find $O/prof_$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  rm -rf $O/prof_$1
done
grep -i "loss16\|loss_finalize" $O/kernel_stats_C2.csv | cut -c1-160
rm -rf $O/prof_tv
