#!/bin/bash
# round 6, session 2: the threshold path on a matrix in LDS (mid-size kernel, 33 ... 64 unknowns) -- parity and first rates
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6j; mkdir -p $out
export TMPDIR=/tmp
echo "=== mid-size parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mid_size" < /dev/null 2>&1 | tail -5
echo "=== rates"; for c in "nn_34" "nn_48" "nn_64"; do timeout 600 python tools/generic_shape_probe.py 8192 1000 "$c" 2>&1 | tail -1; done
echo "=== rates, registers off (matrix in LDS at 20 / 32 unknowns)"; for c in "nn_20" "nn_32"; do ACME_COOP_REG=0 timeout 600 python tools/generic_shape_probe.py 8192 1000 "$c" 2>&1 | tail -1; done
echo "=== literal at 34"; ACME_COOP_LITERAL=1 timeout 600 python tools/generic_shape_probe.py 8192 400 "nn_34" 2>&1 | tail -1
