#!/bin/bash
# final bench lines: the driver's command (with the CPU baseline leg) and the other configurations
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_full.json 2> gpurun_out/$tag/bench_full.err
tail -1 gpurun_out/$tag/bench_full.json | cut -c1-300
bash tools/gpu_final_benches.sh $tag
