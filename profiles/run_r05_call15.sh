#!/bin/bash
# Round 5, GPU call 15: the whole GPU suite on the round's last library build; the multi-rank code path of bench.py end to end
# with 2 and 4 ranks on ONE GPU over the host data plane (functional, not a performance number) and with one rank over RCCL.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for n in 2 4; do
  VAMBHIP_BENCH_HOST_PLANE=1 timeout 600 python bench.py --gpus $n --config C1 --epochs 20 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-taxvamb > $O/bench_${n}ranks_c1_host_plane.json 2> $O/bench_${n}ranks_c1.err
  tail -c 300 $O/bench_${n}ranks_c1_host_plane.json; echo
done
VAMBHIP_BENCH_HOST_PLANE=1 timeout 600 python bench.py --gpus 2 --contigs 300000 --samples 200 --batch 8192 --dtype bf16 --epochs 30 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-taxvamb > $O/bench_2ranks_bf16_host_plane.json 2> $O/bench_2ranks_bf16.err
tail -c 300 $O/bench_2ranks_bf16_host_plane.json; echo
timeout 600 python bench.py --gpus 1 --force-dist --contigs 300000 --samples 200 --batch 8192 --dtype bf16 --epochs 30 --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --no-taxvamb > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err
tail -c 300 $O/bench_1rank_rccl.json; echo; tail -3 $O/bench_1rank_rccl.err
