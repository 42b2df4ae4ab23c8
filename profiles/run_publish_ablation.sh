R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for d in 16 32 48; do
  rm -rf /tmp/pp
  VAMBHIP_SCAN_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python $R/tools/gpu/gpu_cluster_blob.py 200000 > /dev/null 2>&1
  f=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
  echo "dbg=$d"; grep -E "clu_publish_kernel|clu_scan_mfma" $f | sed 's/(float const.*)",/",/; s/(int, unsigned.*)",/",/' | cut -c1-150
done
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python $R/tools/gpu/gpu_cluster_blob.py 200000 > /dev/null 2>&1
f=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
echo "dbg=0"; grep -E "clu_publish_kernel|clu_scan_mfma" $f | sed 's/(float const.*)",/",/; s/(int, unsigned.*)",/",/' | cut -c1-150
