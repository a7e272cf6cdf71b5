#!/usr/bin/env python3
"""Basic-block view of one kernel of a -save-temps gfx950 assembly file (developer tool).

  tools/cfg.py <file.s> <kernel-substring>            block table: size, instruction mix, successors, loop depth
  tools/cfg.py <file.s> <kernel-substring> dump A B   instructions of blocks A..B

Blocks are label / branch delimited; a branch to an earlier block is a back edge, and the blocks it
spans form a loop (nesting depth = number of enclosing back-edge spans)."""
import collections
import re
import sys


def classify(line):
    op = line.split()[0]
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'rwlane'
    if 'dpp' in line and op.startswith('v_'): return 'dpp'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_cmp'): return 'vcmp'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'vmov'
    if op.startswith('v_'): return 'valu'
    if op == 's_nop': return 'nop'
    if op == 's_waitcnt': return 'wait'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'): return 'vmem'
    return 'other'


def parse(path, kernel):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and kernel in l.split(':')[0])
    blocks = []          # dict(label, instrs)
    cur = dict(labels=['entry'], ins=[])
    for l in lines[start + 1:]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            if cur['ins'] or cur['labels'] == ['entry']:
                blocks.append(cur)
                cur = dict(labels=[m.group(1)], ins=[])
            else:
                cur['labels'].append(m.group(1))
            continue
        if not re.match(r'^\s+[a-z_0-9]+', l) or l.strip().startswith('.') or l.strip().startswith(';'):
            if l.startswith('.Lfunc_end'):
                break
            continue
        ins = l.strip()
        cur['ins'].append(ins)
        op = ins.split()[0]
        if op.startswith('s_cbranch') or op == 's_branch' or op == 's_endpgm' or op.startswith('s_setpc'):
            blocks.append(cur)
            cur = dict(labels=[], ins=[])
    if cur['ins']:
        blocks.append(cur)
    lab2b = {}
    for i, b in enumerate(blocks):
        for lb in b['labels']:
            lab2b[lb] = i
    for i, b in enumerate(blocks):
        succ = []
        last = b['ins'][-1] if b['ins'] else ''
        op = last.split()[0] if last else ''
        if op == 's_branch':
            succ = [lab2b.get(last.split()[1], -1)]
        elif op.startswith('s_cbranch'):
            succ = [lab2b.get(last.split()[1], -1), i + 1]
        elif op == 's_endpgm':
            succ = []
        else:
            succ = [i + 1]
        b['succ'] = succ
    return blocks


def loops(blocks):
    spans = []
    for i, b in enumerate(blocks):
        for s in b['succ']:
            if 0 <= s <= i:
                spans.append((s, i))
    return spans


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    blocks = parse(path, kernel)
    spans = loops(blocks)
    if len(sys.argv) > 3 and sys.argv[3] == 'dump':
        a, b = int(sys.argv[4]), int(sys.argv[5])
        for i in range(a, b + 1):
            print(f"--- block {i} {' '.join(blocks[i]['labels'])} -> {blocks[i]['succ']}")
            for x in blocks[i]['ins']:
                print('   ', x.split(';')[0].rstrip())
        return
    tot = collections.Counter()
    for i, b in enumerate(blocks):
        c = collections.Counter(classify(x) for x in b['ins'])
        tot.update(c)
        depth = sum(1 for s, e in spans if s <= i <= e)
        heads = [f"<{e}" for s, e in spans if s == i]
        mix = ' '.join(f"{k}={v}" for k, v in sorted(c.items()))
        print(f"{i:4d} d{depth} n={len(b['ins']):4d} -> {b['succ']} {' '.join(heads)}  {mix}")
    print("total", dict(tot))
    print("loops (head, latch):", sorted(spans))


if __name__ == '__main__':
    main()
