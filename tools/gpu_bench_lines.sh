#!/bin/bash
# the bench lines of every BASELINE configuration (driver command first) with the committed profiles/pmc_traffic.json
cd $GRAFT_REPO_ROOT
tag=${1:-lines}; mkdir -p gpurun_out/$tag
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_full.json 2> gpurun_out/$tag/bench_full.err
tail -1 gpurun_out/$tag/bench_full.json | cut -c1-220
bash tools/gpu_final_benches.sh $tag
