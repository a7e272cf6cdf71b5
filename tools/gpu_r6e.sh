#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6e; mkdir -p $out
export TMPDIR=/tmp
for lib in "" build_variants/libacme_hip_nomirror.so; do
  echo "=== literal-path probe, lib: ${lib:-product}"
  ACME_HIP_LIB=${lib:+$PWD/$lib} timeout 900 python tools/coop_lit_probe.py 27 2>&1 | grep -v amdgpu.ids | tee -a $out/coop_lit_probe.txt
done
