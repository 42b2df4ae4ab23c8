#!/bin/bash
# Round 5, the round's LAST library build (+ the speculative fill one pass ahead): smoke, whole GPU suite, bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05zzz; mkdir -p $O; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
SECONDS=0
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $SECONDS s" | tee $O/bench_wall.txt; tail -c 300 $O/bench.json
