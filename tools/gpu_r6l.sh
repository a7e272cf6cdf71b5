#!/bin/bash
# round 6, session 2: the small shapes with their rows' constants hoisted into registers (Shape::HOISTROW): configs 5 / 2 and the headline
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6l; mkdir -p $out
export TMPDIR=/tmp
for w in birdie_grid diodeclipper_sweep superover_grid; do
  timeout 900 python bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-host-path --no-other-workloads --no-literal-grid 2> $out/$w.err | tail -1 > $out/$w.json
  python - $out/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["config"]["workload"], "value %.4e" % d["value"], "ms_per_step %.2f" % d["ms_per_step"], "kernel_ms", d.get("kernel_ms"), "y_abs_sum", d.get("y_abs_sum_rank0"), "iters", d["config"].get("newton_iters_per_sample"))
PY
done
