#!/bin/bash
# round 6: the driver's bench command with the new legs, the saturation curve, and the rocprofv3 passes of the mid-size workload
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6f; mkdir -p $out
export TMPDIR=/tmp
echo "=== bench (driver's command)"; (time timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err); tail -c 1500 $out/bench_default.json; tail -3 $out/bench_default.err
echo "=== saturation"; (time timeout 1500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-path --no-other-workloads --saturation > $out/bench_saturation.json 2> $out/bench_saturation.err); python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6f/bench_saturation.json").read().strip().splitlines()[-1])
    for r in d["config"]["saturation"]: print(r)
except Exception as e: print("saturation failed", e)
PY
echo "=== profile clipper_chain_20"; STEPS=3 WARMUP=3 bash tools/profile_gpu.sh r6_clipper_chain_20 --workload clipper_chain_20 2>&1 | tail -40
