#!/bin/bash
# Build kernel variants (extra -D flags) next to the product library and, on a GPU box, bench each.
#   tools/variants.sh build  "name1:-DFLAG_A" "name2:-DFLAG_A -DFLAG_B" ...
#   tools/variants.sh bench  [bench.py args]        (runs every build_variants/*.so)
# A variant's translation units compile in parallel (the kernels are split over acme_hip_part<k>.hip).
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p build_variants
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    tmp=$(mktemp -d)
    for tu in acme_hip acme_hip_part0 acme_hip_part1 acme_hip_part2 acme_hip_part3 acme_hip_part4 acme_hip_part5 acme_hip_part6 acme_hip_part7; do
      unit=""      # the product's per-unit flags (__graft_entry__.py: HIP_UNIT_FLAGS); NOUNIT=1 builds every unit alike
      case $tu in acme_hip_part[12345]) [ -z "$NOUNIT" ] && unit="-mllvm -amdgpu-sched-strategy=${UNITSTRAT:-max-ilp}";; esac
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -join-splitedges=1 $unit $flags -c acme_jl_amd/csrc/$tu.hip -o $tmp/$tu.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o build_variants/libacme_hip_$name.so
    rm -rf $tmp
  done
else
  for so in build_variants/*.so; do
    echo "== $so"
    ACME_HIP_LIB=$PWD/$so timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1
  done
fi
