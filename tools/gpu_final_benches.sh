#!/bin/bash
# bench lines of every BASELINE configuration with the current library -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
run() { name=$1; shift; timeout 200 python bench.py --no-cpu-baseline --no-other-workloads "$@" > gpurun_out/$tag/bench_$name.json 2> gpurun_out/$tag/bench_$name.err; python - gpurun_out/$tag/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4g inst*samples/s  %.1f ms/step  kernel %.1f ms  its %.3f warn %g" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["newton_iters_per_sample"], d["config"]["n_warn"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run diodeclipper_sweep --workload diodeclipper_sweep --steps 5 --warmup 2
ACME_LANE_KERNEL=0 run diodeclipper_sweep_16lane --workload diodeclipper_sweep --steps 5 --warmup 2
run birdie_grid --workload birdie_grid --steps 3 --warmup 1
run montecarlo_T44100 --workload superover_montecarlo --steps 3 --warmup 2
run full_homotopy --solver homotopy --steps 3 --warmup 2
run full_gather_rccl1 --gather rank0 --steps 2 --warmup 1
