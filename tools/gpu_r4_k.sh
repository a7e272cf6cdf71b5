#!/bin/bash
# instances per wave (ACME_WAVE_DENSITY) on the configurations that do not fill the chip.   usage: tools/gpu_r4_k.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 300 python -m pytest tests/test_gpu_headline.py -x -q -k "balance" 2>&1 | tail -4
line() { python -c "import sys,json; d=json.loads(open('gpurun_out/$tag/b.json').read()); print('$1', '%.4g' % d['value'], '%.1f ms' % d['ms_per_step'], d['config'].get('y_abs_sum_rank0'), d['config']['newton_iters_per_sample'])"; }
for dens in 4 2 1 auto; do
  if [ $dens = auto ]; then unset ACME_WAVE_DENSITY; else export ACME_WAVE_DENSITY=$dens; fi
  timeout 120 python bench.py --no-cpu-baseline --workload birdie_grid --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/$tag/b.json; line "birdie_grid density $dens"
  ACME_LANE_KERNEL=0 timeout 120 python bench.py --no-cpu-baseline --workload diodeclipper_sweep --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/$tag/b.json; line "diodeclipper 16-lane density $dens"
done
