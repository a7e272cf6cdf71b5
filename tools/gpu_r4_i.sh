#!/bin/bash
# placement of the waves by cost: the GPU test, then the headline with and without it.   usage: tools/gpu_r4_i.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 300 python -m pytest tests/test_gpu_headline.py -x -q -k "balance or isolation or pathological" 2>&1 | tail -4
for bal in 0 1 auto; do
  if [ $bal = auto ]; then unset ACME_BALANCE; else export ACME_BALANCE=$bal; fi
  timeout 120 python bench.py --no-cpu-baseline --no-host-path --steps 10 --warmup 5 2>&1 | tail -1 > gpurun_out/$tag/bench_balance_$bal.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/$tag/bench_balance_$bal.json').read()); print('balance $bal', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('y_abs_sum_rank0'), d['config']['newton_iters_per_sample'])"
done
