#!/bin/bash
# round-end evidence in one GPU call: the parity suite, the driver's bench command, the other configurations'
# bench lines, rocprofv3 passes of the headline and of configs 4, 5, 2.   usage: tools/gpu_round_end.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/$tag/pytest_gpu.txt
bash tools/profile_gpu.sh ${tag}_hl --steps 3 --warmup 2 > gpurun_out/$tag/profile_hl.txt 2>&1
bash tools/profile_gpu.sh ${tag}_mc --workload superover_montecarlo --steps 2 --warmup 1 > gpurun_out/$tag/profile_mc.txt 2>&1
bash tools/profile_gpu.sh ${tag}_birdie --workload birdie_grid --steps 2 --warmup 1 > gpurun_out/$tag/profile_birdie.txt 2>&1
bash tools/profile_gpu.sh ${tag}_diode --workload diodeclipper_sweep --steps 2 --warmup 1 > gpurun_out/$tag/profile_diode.txt 2>&1
bash tools/profile_mix.sh ${tag}_hl --steps 2 --warmup 1 > /dev/null 2>&1
bash tools/profile_mix.sh ${tag}_mc --workload superover_montecarlo --steps 2 --warmup 1 > /dev/null 2>&1
grep -h "dispatches=3\|dispatches=5" gpurun_out/$tag/profile_hl.txt | head -30
