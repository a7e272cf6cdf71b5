#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
export ACME_HIP_LIB=$PWD/build_variants/t/libacme_hip_timing.so ACME_LANE_KERNEL=0
ACME_PROBE_WORKLOAD=birdie_grid ACME_PROBE_SOLVER=homotopy timeout 120 python tools/timing_probe.py 8820 2048 2>&1 | tee gpurun_out/$1/probe_birdie.txt
ACME_PROBE_WORKLOAD=diodeclipper_sweep timeout 120 python tools/timing_probe.py 4410 4096 2>&1 | tee gpurun_out/$1/probe_diode.txt
