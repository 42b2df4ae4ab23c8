cd $GRAFT_REPO_ROOT
for d in 1 2; do
  VAMBHIP_SCAN_DBG=$d timeout 200 python tests/gpu_scan_bench.py gpurun_out/r02d/scan_bench_dbg$d.json 2>&1 | grep -E "n=(100000|2000000) L=32 k=( 1| 8|16|32)" | sed "s/^/dbg=$d /"
done
