#!/bin/bash
# Round 4: the driver's bench command with the host-buffer leg, and the GPU suite.
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/$tag/bench_host.json 2> gpurun_out/$tag/bench_host.err
tail -1 gpurun_out/$tag/bench_host.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['config']['host_buffers'], indent=1))"
tail -3 gpurun_out/$tag/bench_host.err

