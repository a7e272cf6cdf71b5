#!/bin/bash
# round-end evidence in one GPU call: the driver's bench command, the other configurations' bench lines,
# rocprofv3 passes of configs 4 and 2.   usage: tools/gpu_round_end.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_full.json 2> gpurun_out/$tag/bench_full.err
tail -1 gpurun_out/$tag/bench_full.json | cut -c1-400
bash tools/gpu_final_benches.sh $tag
bash tools/profile_gpu.sh ${tag}_mc --workload superover_montecarlo --steps 2 --warmup 1 > gpurun_out/$tag/profile_mc.txt 2>&1
bash tools/profile_gpu.sh ${tag}_diode --workload diodeclipper_sweep --steps 2 --warmup 1 > gpurun_out/$tag/profile_diode.txt 2>&1
tail -40 gpurun_out/$tag/profile_mc.txt
