#!/bin/bash
# bench every library under build_variants/ on the headline workload.   usage: tools/gpu_r4_c.sh <tag> [bench args]
cd $GRAFT_REPO_ROOT
tag=$1; shift; mkdir -p gpurun_out/$tag
for so in build_variants/*.so; do
  echo "== $so"
  ACME_HIP_LIB=$PWD/$so timeout 90 python bench.py --no-cpu-baseline --steps 5 --warmup 3 "$@" 2>&1 | tail -1 | tee gpurun_out/$tag/bench_$(basename $so .so).json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('y_abs_sum_rank0'), d['config'].get('iters_per_sample'))"
done
