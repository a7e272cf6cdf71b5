#!/bin/bash
# A/B of build_variants/*.so on the small-shape workloads, one repetition, config 4 at its stated length.
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
export ACME_LANE_KERNEL=${LANE:-0}
for wl in ${@:-superover_montecarlo birdie_grid diodeclipper_sweep}; do
  for so in build_variants/*.so; do
    r=$(ACME_HIP_LIB=$PWD/$so timeout ${BENCH_TIMEOUT:-120} python bench.py --no-cpu-baseline --no-host-path --no-other-workloads --workload $wl --steps ${STEPS:-3} --warmup ${WARMUP:-2} 2>&1 | tail -1)
    echo "$r" >> gpurun_out/$tag/ab_${wl}.jsonl
    echo "$wl $(basename $so) $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g inst*samples/s  %.1f ms/step  its %.3f  checksum %.12g" % (d["value"], d["ms_per_step"], d["config"]["newton_iters_per_sample"], d["config"]["y_abs_sum_rank0"]))' 2>&1 | tail -1)"
  done
done
