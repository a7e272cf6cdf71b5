#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6d; mkdir -p $out
export TMPDIR=/tmp
echo "=== mid-size parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "mid_size" < /dev/null > $out/pytest_mid_size_full.txt 2>&1; grep -v "^  File" $out/pytest_mid_size_full.txt | grep "mid-size kernel,\|Error\|fault\|passed\|failed" | cut -c1-400
echo "=== throughput"
timeout 600 python tools/generic_shape_probe.py 8192 1102 "clipper chain, 1" < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/shape_probe.txt
