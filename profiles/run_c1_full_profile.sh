R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o trace -- python $R/bench.py --config C1 --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 > $O/bench_c1_under_rocprof.json 2> $O/prof.err
f=$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_c1_full.csv
head -12 $O/kernel_stats_bench_c1_full.csv | cut -c1-160
grep -E "clu_|prep_" $O/kernel_stats_bench_c1_full.csv | cut -c1-200
cut -c1-300 $O/bench_c1_under_rocprof.json
