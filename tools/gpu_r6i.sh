#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6i; mkdir -p $out
export TMPDIR=/tmp
echo "=== const rows test"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "constant_input_rows" < /dev/null 2>&1 | tail -3
echo "=== bench host leg"; timeout 600 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-workloads > $out/bench_host.json 2> $out/bench_host.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6i/bench_host.json").read().strip().splitlines()[-1])
h = d["config"]["host_buffers"]
print("ms_per_step", d["ms_per_step"], {k: v for k, v in h.items() if k != "note"})
PY
