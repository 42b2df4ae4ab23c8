cd $GRAFT_REPO_ROOT
for d in 0 4 8 12 1; do
  VAMBHIP_SCAN_DBG=$d timeout 200 python tools/gpu/gpu_scan_bench.py 2>&1 | grep -E "n=(100000|400000|2000000) L=32 k=( 1| 8|25)" | sed "s/^/dbg=$d /"
done
