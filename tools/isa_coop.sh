#!/bin/bash
# Static look at the mid-size kernel's register instantiations (no GPU needed): builds acme_hip_coop<NC>.hip alone with
# -save-temps into /tmp/isa_coop<NC>_<tag> and prints registers, spills and the instruction mix of every kernel.
#   usage: tools/isa_coop.sh <NC> <tag> [extra hipcc flags]
set -e
nc=${1:-20}; tag=${2:-dev}; shift 2 || true
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/isa_coop${nc}_$tag
mkdir -p $out && cd $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -join-splitedges=1 -c -save-temps "$@" \
    $root/acme_jl_amd/csrc/acme_hip_coop$nc.hip -o $out/unit.o 2> $out/build.log || { tail -30 $out/build.log; exit 1; }
S=acme_hip_coop$nc-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" $S | paste - - - - - - - | sed 's/\s\+/ /g'
python3 - $S <<'PY'
import re, sys, collections
cur=None; stats={}
for l in open(sys.argv[1]):
    m=re.match(r'^(_Z\w+):', l)
    if m: cur=m.group(1); stats[cur]=collections.Counter(); continue
    if cur is None or not re.match(r'^\s+[a-z_0-9]+ ', l): continue
    op=l.split()[0]; c=stats[cur]
    if op.startswith(('v_readlane','v_writelane')): c['rwlane']+=1
    elif op.startswith('v_accvgpr'): c['acc']+=1
    elif op.startswith('v_'): c['valu']+=1
    elif op.startswith('s_nop'): c['s_nop']+=1
    elif op.startswith('s_waitcnt'): c['s_waitcnt']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith('scratch_'): c['scratch']+=1
    elif op.startswith(('global_','buffer_')): c['global']+=1
    if op=='s_endpgm': cur=None
for k,c in stats.items():
    if c: print(k, sum(c.values()), dict(c))
PY
