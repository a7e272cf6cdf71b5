#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6d; mkdir -p $out
export TMPDIR=/tmp
echo "=== mid-size parity"; timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "mid_size" < /dev/null > $out/pytest_mid_size_full.txt 2>&1; grep -v "^  File" $out/pytest_mid_size_full.txt | head -40 | cut -c1-400; tail -5 $out/pytest_mid_size_full.txt
for lit in 0 1; do
  echo "=== throughput, ACME_COOP_LITERAL=$lit"
  ACME_COOP_LITERAL=$lit timeout 600 python tools/generic_shape_probe.py 8192 1102 "clipper chain, 1" < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/shape_probe_lit$lit.txt
done
echo "=== birdie gap"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "birdie_var_iteration_gap" < /dev/null 2>&1 | tail -8 | tee $out/pytest_birdie_gap.txt
