#!/bin/bash
# Round 6, GPU call 27: width of the publish kernel behind many-medoid passes (scan.publish_threads_big = 256 / 512 / 1024): phase
# stamps of the publish kernel at three shapes, then C2 sweeps under the three settings in one process (same latents)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y3; mkdir -p $O; cd $R
export VAMBHIP_LIB_PATH=$R/vamb_amd/libvambhip_timing.so
for w in 256 512 1024; do
  for shape in "170000 32 9" "170000 32 16" "620000 32 32"; do
    echo "#### publish threads $w" >> $O/publish_width.txt
    VAMBHIP_SCAN_PUBLISH_THREADS=$w timeout 200 python tools/gpu/gpu_scan_timeline.py $shape >> $O/publish_width.txt 2>&1
  done
done
unset VAMBHIP_LIB_PATH
grep "####\|n=\|flush retired\|publish\|wall" $O/publish_width.txt | cut -c1-150
S=VAMBHIP_SCAN_PUBLISH_THREADS
VAMBHIP_GEN_PROFILE=1 timeout 900 python tools/gpu/gpu_cluster_sweep_ab.py 2000000 200 8192 bf16 300 "$S=256;$S=512;$S=1024;$S=256;$S=512;$S=1024" > $O/sweep_publish_width.txt 2>&1
grep "setting\|generator: total\|with 32 medoids\|with 16 medoids" $O/sweep_publish_width.txt | cut -c1-220
