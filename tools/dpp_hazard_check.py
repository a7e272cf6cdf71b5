#!/usr/bin/env python3
"""Post-build check of the DPP data hazards the compiler cannot see.

gfx950 (like all gfx9) needs 2 wait states between a VALU write of a VGPR and a DPP read of that
VGPR, and 5 between a VALU write of EXEC and any DPP instruction; the hardware does NOT interlock
(tools/ubench/dppfma.hip shows wrong lanes).  The compiler inserts the wait states for DPP
instructions it emits itself, but the kernel's fused v_fmac_f64_dpp / v_mov_b64_dpp come from
inline asm, which its hazard recogniser does not look into (acme_wave_hip.h).  This script
disassembles the gfx950 code object inside a built library and proves, for EVERY DPP instruction
of every kernel and along every path into it (fall-through and branches), that no VALU
instruction within the two preceding wait states writes a register the DPP operand reads, and
that no VALU write of EXEC sits within five.

usage: dpp_hazard_check.py path/to/libacme_hip.so      (exit status 1 on a hazard)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def disassemble(lib):
    """disassembly of every gfx950 code object in the library (one offload bundle per translation unit)"""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib])
        data = open(fat, "rb").read()
        starts = [m for m in range(len(data)) if data.startswith(magic, m)] if data.count(magic) > 1 else [0]
        out = []
        for k, st in enumerate(starts):
            part = os.path.join(d, f"fat{k}.bin")
            with open(part, "wb") as fh:
                fh.write(data[st:starts[k + 1] if k + 1 < len(starts) else len(data)])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}",
                                   f"--input={part}", f"--output={co}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            out.append(subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True))
        return "\n".join(out)


INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]{12}):\s+((?:[0-9A-Fa-f]{8} ?)+)")


def vregs(tok):
    """set of VGPR numbers named by an operand token like v12, v[12:13], -v[4:5], |v3|"""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"(?<![a-z_])v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def parse(text):
    """[(name, [ins...])] per kernel; ins = dict(addr, op, ops (operand tokens), size)"""
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = []
            kernels.append((m.group(1), cur))
            continue
        m = INS.match(line)
        if m and cur is not None:
            ops = [t.strip() for t in m.group(2).split(",")] if m.group(2) else []
            cur.append(dict(addr=int(m.group(3), 16), op=m.group(1), ops=ops, size=4 * len(m.group(4).split()),
                            text=m.group(1) + " " + m.group(2)))
    return kernels


def is_valu(op):
    return op.startswith("v_")


def valu_dst_vregs(ins):
    """VGPRs a VALU instruction writes (first operand, unless it is a compare / readlane writing SGPRs)."""
    op = ins["op"]
    if not is_valu(op) or not ins["ops"]:
        return set()
    if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return set()
    d = vregs(ins["ops"][0])
    if op.startswith("v_swap") or op.startswith("v_permlane"):   # also write their second operand
        d |= vregs(ins["ops"][1])
    return d


def writes_exec_valu(ins):
    return ins["op"].startswith("v_cmpx") or (is_valu(ins["op"]) and ins["ops"] and ins["ops"][0] == "exec")


def wait_states(ins):
    if ins["op"] == "s_nop":
        return int(ins["ops"][0], 0) + 1
    return 1


def check_kernel(name, code):
    idx = {ins["addr"]: i for i, ins in enumerate(code)}
    preds = {}          # index -> [branch instruction indices jumping here]
    for i, ins in enumerate(code):
        if re.match(r"s_c?branch", ins["op"]) and ins["ops"]:
            off = int(ins["ops"][0].split()[0], 0)
            if off >= 32768:
                off -= 65536
            tgt = ins["addr"] + 4 + 4 * off
            if tgt in idx:
                preds.setdefault(idx[tgt], []).append(i)
    problems, ndpp = [], 0

    def walk(i, budget, need, dpp_i, seen):
        """walk backwards from instruction i (exclusive) while `budget` wait states remain"""
        if budget <= 0 or (i, budget) in seen:
            return
        seen.add((i, budget))
        for b in preds.get(i, []):                     # paths arriving by a branch: the branch is a wait state
            visit(b, budget, need, dpp_i, seen)
        if i > 0 and code[i - 1]["op"] not in ("s_branch", "s_endpgm"):
            visit(i - 1, budget, need, dpp_i, seen)

    def visit(j, budget, need, dpp_i, seen):
        ins = code[j]
        if need is EXEC:
            if writes_exec_valu(ins):
                problems.append((code[dpp_i], ins, "VALU write of EXEC"))
        elif valu_dst_vregs(ins) & need:
            problems.append((code[dpp_i], ins, "VALU write of the DPP source"))
        walk(j, budget - wait_states(ins), need, dpp_i, seen)

    EXEC = object()
    for i, ins in enumerate(code):
        if "_dpp" not in ins["op"]:
            continue
        ndpp += 1
        src = vregs(ins["ops"][1])            # DPP applies to src0
        walk(i, 2, src, i, set())
        walk(i, 5, EXEC, i, set())
    return ndpp, problems


def main(lib):
    total, bad = 0, 0
    for name, code in parse(disassemble(lib)):
        n, problems = check_kernel(name, code)
        total += n
        for dpp, w, why in problems:
            bad += 1
            print(f"HAZARD in {name[:60]}: {why}\n    {w['addr']:08x}: {w['text']}\n    {dpp['addr']:08x}: {dpp['text']}")
    print(f"dpp_hazard_check: {total} DPP instructions checked, {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
