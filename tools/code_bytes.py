#!/usr/bin/env python3
"""Code size of the loops of a gfx950 code object (llvm-objdump -d output on stdin or a path):
for every backward branch the instruction count and byte size of the loop body.  Per-wave
instruction fetch is a first-order cost of the run kernel ('what bounds the kernel')."""
import re
import sys

def main(path):
    rows = []
    for l in open(path):
        m = re.match(r'\s+(\S.*?)\s+//\s+([0-9A-F]{12}):\s+((?:[0-9A-F]{8} ?)+)', l)
        if m:
            rows.append((int(m.group(2), 16), m.group(1).strip(), len(m.group(3).split()) * 4))
    print(len(rows), "instructions,", sum(r[2] for r in rows), "bytes")
    addr = {r[0]: i for i, r in enumerate(rows)}
    for i, (a, txt, sz) in enumerate(rows):
        m = re.match(r's_c?branch\w*\s+(\d+)', txt)
        if m and int(m.group(1)) >= 32768:
            tgt = a + 4 + (int(m.group(1)) - 65536) * 4
            if tgt in addr:
                j = addr[tgt]
                n = i - j + 1
                b = sum(r[2] for r in rows[j:i + 1])
                if n > 100:
                    print(f"loop at instr {j}..{i}: {n} instructions, {b} bytes, {b / n:.2f} B/instr")

if __name__ == '__main__':
    main(sys.argv[1])
