#!/bin/bash
# Build kernel variants (extra -D flags) next to the product library and, on a GPU box, bench each.
#   tools/variants.sh build  "name1:-DFLAG_A" "name2:-DFLAG_A -DFLAG_B" ...
#   tools/variants.sh bench  [bench.py args]        (runs every build_variants/*.so)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p build_variants
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
        acme_jl_amd/csrc/acme_hip.hip acme_jl_amd/csrc/acme_hip_part[0-3].hip -o build_variants/libacme_hip_$name.so &
  done
  wait
else
  for so in build_variants/*.so; do
    echo "== $so"
    ACME_HIP_LIB=$PWD/$so timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1
  done
fi
