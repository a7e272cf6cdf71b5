#!/bin/bash
# round 6: the rocprofv3 evidence of every bench workload (kernel trace + PMC passes, steady state), one GPU call.
# Afterwards, locally: tools/merge_pmc.py r6 gpurun_out/prof_r6_*   (profiles/pmc_traffic.json + the summaries under profiles/)
cd $GRAFT_REPO_ROOT
for wl in ${WORKLOADS:-superover_grid superover_montecarlo birdie_grid diodeclipper_sweep clipper_chain_20 clipper_chain_34}; do
  echo "=== profile $wl"
  STEPS=3 WARMUP=3 bash tools/profile_gpu.sh r6_$wl --workload $wl 2>&1 | tail -12
done
