#!/bin/bash
# A/B of the libraries under build_variants/ on the bench workloads, in one GPU call (same box, same clocks).
#   tools/gpu_ab.sh <tag> [workload ...]    default workloads: superover_grid
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
wls=${@:-superover_grid}
for rep in 1 2; do
for wl in $wls; do
  for so in build_variants/*.so; do
    extra=""
    [ "$wl" = superover_montecarlo ] && extra="--samples 8820"
    r=$(ACME_HIP_LIB=$PWD/$so timeout ${BENCH_TIMEOUT:-100} python bench.py --no-cpu-baseline --workload $wl --steps ${STEPS:-3} --warmup ${WARMUP:-1} $extra 2>&1 | tail -1)
    echo "$r" >> gpurun_out/$tag/ab_${wl}.jsonl
    echo "$wl $(basename $so) $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g inst*samples/s  %.1f ms/step  its %.3f  checksum %.12g" % (d["value"], d["ms_per_step"], d["config"]["newton_iters_per_sample"], d["config"]["y_abs_sum_rank0"]))' 2>&1 | tail -1)"
  done
done
done
