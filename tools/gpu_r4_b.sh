#!/bin/bash
# Round 4: steady-state SQ counters of the libraries under build_variants/ and the in-situ timing breakdown of the
# timing build (build_timing/).   usage: tools/gpu_r4_b.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; mkdir -p gpurun_out/$tag
STEPS=3 WARMUP=3 bash tools/pmc_ab.sh $tag > gpurun_out/$tag/pmc.log 2>&1; tail -60 gpurun_out/$tag/pmc_ab.txt
for so in build_timing/*.so; do
  echo "== timing $so"
  ACME_HIP_LIB=$PWD/$so timeout 200 python tools/timing_probe.py 2205 8192 2>&1 | tail -18 | tee gpurun_out/$tag/timing_$(basename $so .so).txt
done
