#!/bin/bash
# round 6, end: the whole GPU suite, the driver's bench command, the other configurations' bench lines, the saturation curve,
# the mid-size sizes -- with profiles/pmc_traffic.json as committed.  Copy what is to be judged from gpurun_out/r6end into profiles/.
cd $GRAFT_REPO_ROOT
tag=r6end; mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q -s 2>&1 | tee gpurun_out/$tag/pytest_gpu_full.txt | tail -3
(time python bench.py --steps 20 --warmup 5) > gpurun_out/$tag/bench_headline_driver_command.json 2> gpurun_out/$tag/bench_headline_driver_command.err
tail -1 gpurun_out/$tag/bench_headline_driver_command.json | cut -c1-260; tail -4 gpurun_out/$tag/bench_headline_driver_command.err
bash tools/gpu_final_benches.sh $tag
timeout 1500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-path --no-other-workloads --saturation > gpurun_out/$tag/bench_saturation.json 2> gpurun_out/$tag/bench_saturation.err
python - <<'PY' | tee gpurun_out/r6end/saturation.txt
import json
d = json.loads([l for l in open("gpurun_out/r6end/bench_saturation.json") if l.startswith("{")][-1])
print("# bench.py --saturation: every workload at 1 / 2 / 4 / 8 x its BASELINE per-GPU size on ONE MI355X (instance*samples/s; never `value`)")
for r in d["config"]["saturation"]:
    if "error" in r: print(r["workload"], r["factor"], "ERROR", r["error"]); continue
    print("%-22s x%d  %6d instances  %.4e  (%.1f ms per step of %d samples, %.2f its/sample, waves per SIMD %s)" % (r["workload"], r["factor"], r["instances"], r["value"], r["ms_per_step"], r["samples_per_step"], r["newton_iters_per_sample"], r["waves_per_simd"]))
PY
for c in nn_4_ nn_8_ nn_16 nsub_4 nsub_6 nsub_8 nsub_9 nn_20 nn_24 nn_32 nn_34 nn_48 nn_64; do timeout 300 python tools/generic_shape_probe.py 8192 2205 "$c" 2>&1 | tail -1; done | tee gpurun_out/$tag/generic_shape_probe.txt
