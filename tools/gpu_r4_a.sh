#!/bin/bash
# Round 4, first GPU call: parity suite, the headline bench with the condensed kernel (default), the plain
# kernel (ACME_CONDENSE=0) on the same box, and the build variants.   usage: tools/gpu_r4_a.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/$tag/pytest_gpu.txt
echo "== condensed (default)"
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 3 2>gpurun_out/$tag/bench_cond.err | tail -1 | tee gpurun_out/$tag/bench_cond.json | cut -c1-400
echo "== plain (ACME_CONDENSE=0)"
ACME_CONDENSE=0 timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 3 2>gpurun_out/$tag/bench_plain.err | tail -1 | tee gpurun_out/$tag/bench_plain.json | cut -c1-400
for so in build_variants/*.so; do
  echo "== $so"
  ACME_HIP_LIB=$PWD/$so timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 3 2>&1 | tail -1 | tee gpurun_out/$tag/bench_$(basename $so .so).json | cut -c1-400
done
