#!/bin/bash
# Build the product library of another git revision as a bench variant:
#   tools/build_ref.sh <name> <git-ref> [extra hipcc flags]   -> build_variants/libacme_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; ref=$2; shift 2
tmp=$(mktemp -d)
git archive "$ref" acme_jl_amd/csrc include | tar -x -C "$tmp"
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" \
    "$tmp/acme_jl_amd/csrc/acme_hip.hip" -o build_variants/libacme_hip_$name.so
rm -rf "$tmp"
