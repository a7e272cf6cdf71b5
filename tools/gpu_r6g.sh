#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6g; mkdir -p $out
export TMPDIR=/tmp
echo "=== throughput, beyond 32 unknowns"
for c in "nn_34" "nn_48" "nn_64"; do timeout 900 python tools/generic_shape_probe.py 8192 441 "$c" < /dev/null 2>&1 | grep -v amdgpu.ids | tee -a $out/shape_probe_big.txt; done
echo "=== GPU tests: per-instance elements"; timeout 900 python -m pytest tests/test_gpu_longrun.py -m gpu -x -q -s -k "per_instance_element" < /dev/null 2>&1 | tail -6
