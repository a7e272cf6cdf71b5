#!/bin/bash
# streamed host run: where the time goes (copies vs kernel), chunk sizes.   usage: tools/gpu_r4_m.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "host_buffer" 2>&1 | tail -2
for ch in 256 128 512; do
ACME_HOST_STREAM_CHUNK=$ch ACME_HOST_STREAM_DEBUG=1 timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 2 2> gpurun_out/$tag/err.txt | tail -1 > gpurun_out/$tag/bench_host.json
echo chunk $ch; grep streamed gpurun_out/$tag/err.txt | head -3
python -c "
import json; d=json.loads(open('gpurun_out/$tag/bench_host.json').read()); hb=d['config']['host_buffers']
print('device', round(d['ms_per_step'],1), {k: round(v,1) for k,v in hb.items() if k.endswith('_ms')})"
done
