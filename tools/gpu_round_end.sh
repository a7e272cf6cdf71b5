#!/bin/bash
# round-end evidence in one GPU call: the parity suite, the driver's bench command, the other configurations'
# bench lines, rocprofv3 passes (headline, configs 4 and 5) and the instruction-mix counters.
# usage: tools/gpu_round_end.sh <tag>     (the bench lines read profiles/pmc_traffic.json as committed)
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/$tag/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_full.json 2> gpurun_out/$tag/bench_full.err
tail -1 gpurun_out/$tag/bench_full.json | cut -c1-260
bash tools/gpu_final_benches.sh $tag
bash tools/profile_gpu.sh ${tag}_hl --steps 3 --warmup 2 > gpurun_out/$tag/profile_hl.txt 2>&1
bash tools/profile_gpu.sh ${tag}_mc --workload superover_montecarlo --steps 2 --warmup 1 > gpurun_out/$tag/profile_mc.txt 2>&1
bash tools/profile_gpu.sh ${tag}_birdie --workload birdie_grid --steps 2 --warmup 1 > gpurun_out/$tag/profile_birdie.txt 2>&1
bash tools/profile_mix.sh ${tag}_hl --steps 2 --warmup 1 > /dev/null 2>&1
bash tools/profile_mix.sh ${tag}_mc --workload superover_montecarlo --steps 2 --warmup 1 > /dev/null 2>&1
