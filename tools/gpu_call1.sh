#!/bin/bash
# round-2 GPU call 1: new parity tests (calibration prints), baseline bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_headline.py tests/test_gpu_pins.py -m gpu -x -q -s > gpurun_out/r2a/pytest_new.log 2>&1
echo "new tests rc=$?"
python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_headline.py > gpurun_out/r2a/pytest_all.log 2>&1
echo "all tests rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_head.json 2> gpurun_out/r2a/bench_head.err
python bench.py --workload superover_montecarlo --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_mc_T44100.json 2> gpurun_out/r2a/bench_mc.err
tail -c 600 gpurun_out/r2a/bench_head.json; tail -c 900 gpurun_out/r2a/bench_mc_T44100.json
