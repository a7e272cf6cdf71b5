#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_r4_c.sh $1 --no-host-path
for d in 64 16 8 4; do
  echo "== diode clipper sweep, lane density $d"
  ACME_LANE_DENSITY=$d timeout 90 python bench.py --no-cpu-baseline --no-host-path --workload diodeclipper_sweep --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('y_abs_sum_rank0'))"
done
