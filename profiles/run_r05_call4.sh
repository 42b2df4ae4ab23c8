#!/bin/bash
# Round 5, GPU call 4: SQ counters of the matrix-pipe scan kernel (32 medoids, 620 k rows) with everything on and with the hit
# path / flush off (scan.debug 13); more two-stream schedules of the step; the 8-rank native sharded sweep on one GPU.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_vae_gpu.py::test_step_scheduling_variants_are_bit_identical -m gpu -q --maxfail=4 > $O/pytest_parallel.log 2>&1; tail -3 $O/pytest_parallel.log
P="VAMBHIP_VAE_FORK_PLAN"; F="VAMBHIP_VAE_FORK_AT_LOSS=1"
timeout 600 python tools/gpu/gpu_step_ab.py 2000000 200 8192 10 bf16 "|$F;$P=6|$F;$P=2|$F;$P=4|$F;$P=10|$F;$P=14|$P=10" 2 > $O/step_fork_plans2_c2.txt 2>&1; grep SUMMARY $O/step_fork_plans2_c2.txt
timeout 400 python tools/gpu/gpu_step_ab.py 2000000 1000 8192 5 bf16 "|$F;$P=6" 2 > $O/step_fork_plans_c3.txt 2>&1; grep SUMMARY $O/step_fork_plans_c3.txt
cd /tmp && export TMPDIR=/tmp
SET1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SET2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES"
SET3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
for dbg in 0 13; do
  for s in 1 2 3; do
    eval "C=\$SET$s"
    rm -rf /tmp/pmcx
    VAMBHIP_SCAN_DBG=$dbg timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/gpu/gpu_scan_one.py 620000 32 32 20 > $O/pmc_scan_dbg${dbg}_set$s.out 2>&1
    f=$(find /tmp/pmcx -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $R/tools/gpu/gpu_pmc_summary.py $f clu_scan_mfma > $O/pmc_scan_mfma_k32_dbg${dbg}_set$s.txt 2>&1
    cat $O/pmc_scan_mfma_k32_dbg${dbg}_set$s.txt | cut -c1-420
  done
done
