#!/bin/bash
# host-buffer pipelines: the GPU test, then the bench's host leg with a sweep of the slice count.   usage: tools/gpu_r4_l.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "host_buffer" 2>&1 | tail -4
ACME_BENCH_HOST_SLICES=8,24 timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 3 2>&1 | tail -1 > gpurun_out/$tag/bench_host.json
python -c "
import json; d=json.loads(open('gpurun_out/$tag/bench_host.json').read()); hb=d['config']['host_buffers']
print('device', round(d['ms_per_step'],1), {k: round(v,1) for k,v in hb.items() if k.endswith('_ms')})"
