#!/usr/bin/env python3
"""Static instruction statistics of a -save-temps gfx950 assembly file: per straight-line block
(label/branch delimited) the instruction count and the VALU / DPP / SGPR-spill (v_readlane,
v_writelane) / scratch mix.  Used to keep the solver loop's issue-slot budget in view."""
import collections
import re
import sys

def main(path, lo=0, minsize=40):
    L = [l.rstrip() for l in open(path) if re.match(r'^\s+[a-z_0-9]+ ', l) or re.match(r'^\.LBB', l)]
    start = 0
    for i, l in enumerate(L + ['.LBBend']):
        if l.startswith('.LBB') or 's_cbranch' in l or 's_branch' in l:
            n = i - start
            if n >= minsize and start >= lo:
                c = collections.Counter()
                for x in L[start:i]:
                    op = x.split()[0]
                    if 'dpp' in x: c['dpp'] += 1
                    elif op.startswith('v_readlane') or op.startswith('v_writelane'): c['rwlane'] += 1
                    elif op.startswith('v_cndmask'): c['cndmask'] += 1
                    elif op.startswith('v_cmp'): c['vcmp'] += 1
                    elif op.startswith('v_mov'): c['vmov'] += 1
                    elif op.startswith('v_'): c['valu'] += 1
                    elif op.startswith('s_'): c['salu'] += 1
                    elif op.startswith('ds_'): c['lds'] += 1
                    elif op.startswith('scratch_'): c['scratch'] += 1
                    elif op.startswith('global_'): c['global'] += 1
                print(f"{start:6d}-{i:6d} n={n:4d} " + ' '.join(f"{k}={v}" for k, v in sorted(c.items())))
            start = i + 1

if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
