#!/bin/bash
# round 6, session 2: cycles per region of the mid-size kernel on a matrix in LDS (-DACME_COOP_TIMING build), rates, parity
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6k; mkdir -p $out
export TMPDIR=/tmp
for st in ${STAGES:-17 24 32}; do ACME_HIP_LIB=$PWD/build_variants/libacme_hip_timing.so timeout 600 python tools/coop_timing_probe.py $st 8192 600 2>&1 | grep -v amdgpu.ids | tail -14; done | tee $out/timing.txt
for c in nn_34 nn_48 nn_64; do timeout 600 python tools/generic_shape_probe.py 8192 1000 $c 2>&1 | tail -1; done | tee $out/rates.txt
echo "=== registers off"; for c in nn_20 nn_32; do ACME_COOP_REG=0 timeout 600 python tools/generic_shape_probe.py 8192 1000 $c 2>&1 | tail -1; done | tee -a $out/rates.txt
echo "=== mid-size parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mid_size" < /dev/null 2>&1 | tail -5
