#!/bin/bash
# Round-3 evidence in one GPU call: the parity suite, the driver's bench command, every other configuration's
# bench line, the generic-shape probe, and the steady-state rocprofv3 passes of configurations 3, 4, 5 and 2.
# usage: tools/gpu_round3_end.sh <tag>   ->  gpurun_out/<tag>/ and gpurun_out/prof_<tag>_*/ (tools/merge_pmc.py)
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/$tag/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_headline_driver_command.json 2> gpurun_out/$tag/bench_headline.err
tail -1 gpurun_out/$tag/bench_headline_driver_command.json | cut -c1-200
bash tools/gpu_final_benches.sh $tag
timeout 300 python tools/generic_shape_probe.py > gpurun_out/$tag/generic_shape_probe.txt 2>&1; tail -6 gpurun_out/$tag/generic_shape_probe.txt
STEPS=5 WARMUP=3 bash tools/profile_gpu.sh ${tag}_headline > gpurun_out/$tag/profile_headline.log 2>&1
STEPS=3 WARMUP=2 bash tools/profile_gpu.sh ${tag}_montecarlo --workload superover_montecarlo > gpurun_out/$tag/profile_montecarlo.log 2>&1
STEPS=3 WARMUP=2 bash tools/profile_gpu.sh ${tag}_birdie --workload birdie_grid > gpurun_out/$tag/profile_birdie.log 2>&1
STEPS=3 WARMUP=2 bash tools/profile_gpu.sh ${tag}_diode --workload diodeclipper_sweep > gpurun_out/$tag/profile_diode.log 2>&1
for p in headline montecarlo birdie diode; do tail -2 gpurun_out/$tag/profile_$p.log; done
