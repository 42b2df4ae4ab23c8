#!/bin/bash
# What is run on the GPU box to produce the profiles/rNN_* evidence of a round (gpurun -- bash profiles/run_round.sh TAG [parts]).
# parts: any of  tests bench prof gemm scan pmc   (default: all but pmc)
TAG=${1:-r02a}
PARTS=${2:-"tests bench prof gemm scan"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has tests; then
  (cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
  tail -3 $O/pytest_gpu.log
fi
if has bench; then
  (cd $R && timeout 1500 python bench.py --steps ${BENCH_STEPS:-1} --warmup ${BENCH_WARMUP:-1} > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
  cat $O/bench.json
  tail -3 $O/bench.err
fi
if has prof; then
  rm -rf /tmp/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- \
      python $R/bench.py --epochs 3 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-cluster > $O/bench_under_rocprof.json 2> $O/prof.err
  f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_bench_e3.csv
  t=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python $R/tools/gpu/gpu_timeline16.py $t > $O/step_timeline.txt 2>&1
  head -30 $O/kernel_stats_bench_e3.csv | cut -c1-200
  tail -45 $O/step_timeline.txt
fi
if has gemm; then
  (cd $R && timeout 300 python tools/gpu/gpu_gemm16_bench.py $O/gemm16_bench.json > $O/gemm16_bench.txt 2>&1)
  cat $O/gemm16_bench.txt
fi
if has scan; then
  (cd $R && timeout 300 python tools/gpu/gpu_scan_bench.py $O/scan_bench.json > $O/scan_bench.txt 2>&1)
  cat $O/scan_bench.txt
fi
if has prep; then
  (cd $R && timeout 600 python -m pytest tests/test_prep_gpu.py -m gpu -x -q > $O/pytest_prep.log 2>&1; echo "rc=$?" >> $O/pytest_prep.log)
  tail -15 $O/pytest_prep.log
fi
if has scanab; then      # the scan kernel's column-loop variants: bit-exactness (cluster tests) + duration per medoid count
  for m in ${SCAN_MODES:-1 2 3 4}; do
    (cd $R && VAMBHIP_SCAN_LC=$m timeout 300 python -m pytest tests/test_cluster_gpu.py -m gpu -x -q > $O/pytest_cluster_lc$m.log 2>&1; echo "rc=$?" >> $O/pytest_cluster_lc$m.log)
    tail -2 $O/pytest_cluster_lc$m.log
    (cd $R && VAMBHIP_SCAN_LC=$m timeout 300 python tools/gpu/gpu_scan_bench.py $O/scan_bench_lc$m.json > $O/scan_bench_lc$m.txt 2>&1)
    grep -E "n=(100000|2000000) L=32 k=( 8|12|16|25|32)" $O/scan_bench_lc$m.txt
  done
fi
if has gemmvar; then
  (cd $R && timeout 300 python tools/gpu/gpu_gemm16_variants.py $O/gemm16_variants.json > $O/gemm16_variants.txt 2>&1)
  cat $O/gemm16_variants.txt
fi
if has sweepab; then     # ONE trained C2 latent matrix clustered under several generator / kernel settings
  (cd $R && timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "${SWEEP_SETTINGS:-VAMBHIP_GEN_PROFILE=1,VAMBHIP_SCAN_LC=1;VAMBHIP_GEN_PROFILE=1,VAMBHIP_SCAN_LC=2;VAMBHIP_GEN_PROFILE=1,VAMBHIP_SCAN_LC=3}" $O/sweep_ab.json > $O/sweep_ab.txt 2>&1)
  cat $O/sweep_ab.txt | cut -c1-300
fi
if has pmc; then     # HBM-side traffic of the roofline kernel (K = D encoder GEMM, C2 and C3 shapes) from two --pmc passes
  for shape in "c2 8192 512 320" "c3 8192 512 1120"; do
    set -- $shape; tag=$1; M=$2; N=$3; K=$4
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_$c
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- \
          python $R/tools/gpu/gpu_gemm16_one.py 3 $M $N $K 30 > $O/pmc_${tag}_$c.out 2>&1
    done
    f=$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
    w=$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && [ -n "$w" ] && python $R/tools/gpu/gpu_pmc_traffic.py $f $w "gemm_bf16_kernel<128, 128, 2, 4, 3" $O/pmc_roofline_$tag.json
  done
fi
exit 0
