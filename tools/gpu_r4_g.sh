#!/bin/bash
# Round 4 evidence: GPU suite, the driver's bench command, the other configurations, the steady-state rocprofv3 passes.
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/$tag/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_headline_driver_command.json 2> gpurun_out/$tag/bench_headline.err
tail -1 gpurun_out/$tag/bench_headline_driver_command.json | cut -c1-250
bash tools/gpu_final_benches.sh $tag
STEPS=5 WARMUP=3 timeout 600 bash tools/profile_gpu.sh ${tag}_headline --no-host-path > gpurun_out/$tag/profile_headline.log 2>&1; tail -3 gpurun_out/$tag/profile_headline.log
