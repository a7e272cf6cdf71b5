#!/bin/bash
# round 6, session 2: the driver's bench command (with config.mid_size and the retained constant-rows leg) and the whole GPU suite
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6n; mkdir -p $out
export TMPDIR=/tmp
(time python bench.py --steps 20 --warmup 5) > $out/bench_driver.json 2> $out/bench_driver.err; tail -3 $out/bench_driver.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6n/bench_driver.json") if l.startswith("{")][-1])
print("value %.4e ms %.2f" % (d["value"], d["ms_per_step"]))
for o in d["config"]["other_workloads"]: print(o.get("workload"), "%.3e" % o.get("value", 0), o.get("error"))
for o in d["config"]["mid_size"]: print(o)
h = d["config"]["host_buffers"]; print({k: v for k, v in h.items() if k.startswith("const") or k in ("one_shot_ms", "steady_ms")})
PY
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tee $out/pytest_gpu.txt | tail -4
