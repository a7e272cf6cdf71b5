#!/bin/bash
# A/B of the lane-per-instance kernel against the 16-lane kernel on the small-model workloads
cd $GRAFT_REPO_ROOT
for wl in diodeclipper_sweep birdie_grid; do
  for lk in 0 1; do
    r=$(ACME_LANE_KERNEL=$lk timeout 100 python bench.py --no-cpu-baseline --workload $wl --steps 2 --warmup 1 2>&1 | tail -1)
    echo "$wl lane=$lk $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g inst*samples/s  %.1f ms/step  its %.3f  checksum %.12g warn %g" % (d["value"], d["ms_per_step"], d["config"]["newton_iters_per_sample"], d["config"]["y_abs_sum_rank0"], d["config"]["n_warn"]))' 2>&1 | tail -1)"
  done
done
