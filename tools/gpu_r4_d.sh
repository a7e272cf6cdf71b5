#!/bin/bash
# Round 4: parity suite, headline bench (driver's command), every other configuration, timing breakdown.
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/$tag/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_headline_driver_command.json 2> gpurun_out/$tag/bench_headline.err
tail -1 gpurun_out/$tag/bench_headline_driver_command.json | cut -c1-300
bash tools/gpu_final_benches.sh $tag
for so in build_timing/*.so; do
  echo "== timing $so"
  ACME_HIP_LIB=$PWD/$so timeout 200 python tools/timing_probe.py 2205 8192 2>&1 | tail -16 | tee gpurun_out/$tag/timing_$(basename $so .so).txt
done
