#!/bin/bash
# A build variant of the mid-size kernel only: recompiles acme_hip.hip and the acme_hip_coop<NC>.hip / acme_hip_coopl<NS>.hip units with extra flags
# and links them with the current objects of the other units (csrc/.obj) into build_variants/libacme_hip_<name>.so.
#   usage: tools/variants_coop.sh <name> [hipcc flags]      e.g.  tools/variants_coop.sh nomirror -DACME_COOP_NO_MIRROR
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/acme_jl_amd/csrc; out=/tmp/variants_coop_$name
mkdir -p $out $root/build_variants
for u in acme_hip acme_hip_coop20 acme_hip_coop24 acme_hip_coop28 acme_hip_coop32 acme_hip_coopl1 acme_hip_coopl2 acme_hip_coopl3 acme_hip_coopl4 acme_hip_coopw; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -join-splitedges=1 "$@" -c $src/$u.hip -o $out/$u.o 2> $out/$u.log &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o $src/.obj/acme_hip_part*.o -o $root/build_variants/libacme_hip_$name.so
python3 $root/tools/dpp_hazard_check.py $root/build_variants/libacme_hip_$name.so | tail -1
