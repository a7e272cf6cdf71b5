#!/bin/bash
# placement of the waves by cost along the signal: seconds 1-4 and 5-15 of the headline, with and without.
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
for args in "--steps 3 --warmup 1" "--steps 3 --warmup 1" "--steps 10 --warmup 5"; do
for bal in 0 1; do
  export ACME_BALANCE=$bal
  timeout 120 python bench.py --no-cpu-baseline --no-host-path $args 2>&1 | tail -1 > gpurun_out/$tag/b.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/$tag/b.json').read()); print('$args balance $bal', d['value'], d['ms_per_step'], d['config'].get('y_abs_sum_rank0'), d['config']['newton_iters_per_sample'])"
done; done
