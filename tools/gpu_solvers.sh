#!/bin/bash
# one workload under the three solver stacks (and both kernels where the lane kernel applies)
cd $GRAFT_REPO_ROOT
wl=${1:-diodeclipper_sweep}
for lane in 1 0; do for sv in caching homotopy simple; do
  r=$(ACME_LANE_KERNEL=$lane timeout 120 python bench.py --no-cpu-baseline --workload $wl --solver $sv --steps 3 --warmup 2 2>&1 | tail -1)
  echo "$wl lane=$lane $sv $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g inst*samples/s  %.1f ms/step  its %.3f" % (d["value"], d["ms_per_step"], d["config"]["newton_iters_per_sample"]))' 2>&1 | tail -1)"
done; done
