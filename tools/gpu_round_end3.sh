#!/bin/bash
# headline-only evidence: parity suite, driver's bench command, homotopy / gather lines, rocprofv3 passes + instruction mix
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/$tag/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_full.json 2> gpurun_out/$tag/bench_full.err
tail -1 gpurun_out/$tag/bench_full.json | cut -c1-260
timeout 200 python bench.py --no-cpu-baseline --solver homotopy --steps 3 --warmup 2 > gpurun_out/$tag/bench_full_homotopy.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --gather rank0 --steps 2 --warmup 1 > gpurun_out/$tag/bench_full_gather_rccl1.json 2>/dev/null
bash tools/profile_gpu.sh ${tag}_hl --steps 3 --warmup 2 > gpurun_out/$tag/profile_hl.txt 2>&1
bash tools/profile_mix.sh ${tag}_hl --steps 2 --warmup 1 > /dev/null 2>&1
grep -h "dispatches=5" gpurun_out/$tag/profile_hl.txt | head -30
