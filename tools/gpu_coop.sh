#!/bin/bash
# ONE GPU call for the mid-size cooperative kernel: its parity test, the shape probe with the factors in registers and
# (ACME_COOP_REG=0) in LDS, and -- if build_variants/libacme_hip_cooptiming.so exists -- the in-situ clock.
#   usage (through gpurun): bash tools/gpu_coop.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mid_size" < /dev/null 2>&1 | tail -5 | tee $out/pytest_mid_size.txt
for reg in 1 0; do
  echo "== ACME_COOP_REG=$reg"
  ACME_COOP_REG=$reg timeout 600 python tools/generic_shape_probe.py 8192 1102 "clipper chain, 1" < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/shape_probe_reg$reg.txt
done
for pin in $COOP_PINS; do      # e.g. COOP_PINS="ACME_COOP_IMGL=0 ACME_COOP_WPB=1"
  echo "== $pin"
  env ${pin//,/ } timeout 600 python tools/generic_shape_probe.py 8192 1102 "clipper chain, 1" < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/shape_probe_$pin.txt
done
lib=$PWD/build_variants/libacme_hip_cooptiming.so
if [ -f $lib ]; then
  for st in 10 16; do
    ACME_HIP_LIB=$lib timeout 300 python tools/coop_timing_probe.py $st 8192 1102 < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/coop_timing_$st.txt
  done
fi
