#!/bin/bash
# Static look at the generated gfx950 code (no GPU needed): builds the headline shape only
# (-DACME_DEV_SHAPES) with -save-temps into /tmp/isa_<tag> and prints registers, spills, LDS-independent
# instruction statistics.   usage: tools/isa.sh <tag> [extra hipcc flags]
set -e
tag=${1:-dev}; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/isa_$tag
mkdir -p $out && cd $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -join-splitedges=1 -shared -save-temps -DACME_DEV_SHAPES${DEVSHAPE:+=$DEVSHAPE} "$@" \
    $root/acme_jl_amd/csrc/acme_hip.hip $root/acme_jl_amd/csrc/acme_hip_part[0-7].hip -o $out/lib.so 2> $out/build.log || { tail -30 $out/build.log; exit 1; }
S=acme_hip_part0-hip-amdgcn-amd-amdhsa-gfx950.s      # (developer builds have ONE shape: number 0, i.e. part 0)
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" $S | paste - - - - - - | sed 's/\s\+/ /g'
grep -E '^\s+[a-z_0-9]+ |^\.LBB' $S > code.s
python3 - <<'PY'
import re, collections
L=[l.split()[0] for l in open('code.s') if not l.startswith('.LBB')]
c=collections.Counter()
for op in L:
    if op.startswith('v_readlane') or op.startswith('v_writelane'): c['rwlane']+=1
    elif op.startswith('v_'): c['valu']+=1
    elif op.startswith('s_nop'): c['s_nop']+=1
    elif op.startswith('s_waitcnt'): c['s_waitcnt']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith('scratch_'): c['scratch']+=1
    elif op.startswith('global_'): c['global']+=1
print(len(L), "instructions:", dict(c))
PY
python3 $root/tools/asm_stats.py code.s 0 | sort -t= -k2 -n -r | head -${TOPN:-14}
