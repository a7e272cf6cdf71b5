"""The post-build DPP hazard check (tools/dpp_hazard_check.py): the built product library must be
clean, and the checker itself must see the hazards it is there for (synthetic disassembly)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dpp_hazard_check as chk  # noqa: E402


def _asm(lines, base=0x1000):
    out = ["0000000000001000 <kernel>:"]
    addr = base
    for text, nwords in lines:
        out.append(f"\t{text:60s} // {addr:012X}: " + " ".join(["00000000"] * nwords))
        addr += 4 * nwords
    return "\n".join(out)


def _hazards(lines):
    (name, code), = chk.parse(_asm(lines))
    return chk.check_kernel(name, code)


DPP = ("v_fmac_f64_dpp v[50:51], v[50:51], v[44:45] row_newbcast:1 row_mask:0xf bank_mask:0xf", 2)


def test_checker_flags_a_fresh_dpp_source():
    n, bad = _hazards([("v_mov_b64_e32 v[50:51], v[94:95]", 1), DPP])
    assert n == 1 and len(bad) == 1 and "DPP source" in bad[0][2]
    # one instruction in between is one wait state: still a hazard; two are enough
    assert len(_hazards([("v_mov_b64_e32 v[50:51], v[94:95]", 1), ("s_mov_b32 s5, s4", 1), DPP])[1]) == 1
    assert not _hazards([("v_mov_b64_e32 v[50:51], v[94:95]", 1), ("s_mov_b32 s5, s4", 1), ("s_mov_b32 s6, s4", 1), DPP])[1]
    assert not _hazards([("v_mov_b64_e32 v[50:51], v[94:95]", 1), ("s_nop 1", 1), DPP])[1]
    # half of the register pair is enough; the non-DPP operands are interlocked by the hardware
    assert len(_hazards([("v_mov_b32_e32 v51, v7", 1), DPP])[1]) == 1
    assert not _hazards([("v_mov_b64_e32 v[44:45], v[94:95]", 1), DPP])[1]
    # a DPP instruction is a VALU write itself (chains on its own result)
    assert len(_hazards([DPP, DPP])[1]) == 1


def test_checker_follows_branches_and_exec_writes():
    # the write sits before a branch INTO the block of the DPP instruction (the branch is one wait state)
    lines = [("v_mov_b64_e32 v[50:51], v[94:95]", 1), ("s_branch 2", 1), ("s_nop 0", 1), ("s_nop 0", 1), DPP]
    assert len(_hazards(lines)[1]) == 1
    assert not _hazards([("v_mov_b64_e32 v[50:51], v[94:95]", 1), ("s_nop 0", 1), ("s_branch 0", 1), DPP])[1]
    # VALU write of EXEC needs five wait states before any DPP; SALU writes of EXEC need none
    assert len(_hazards([("v_cmpx_gt_f64_e32 exec, v[2:3], v[4:5]", 1), ("s_nop 3", 1), DPP])[1]) == 1
    assert not _hazards([("v_cmpx_gt_f64_e32 exec, v[2:3], v[4:5]", 1), ("s_nop 4", 1), DPP])[1]
    assert not _hazards([("s_mov_b64 exec, s[4:5]", 1), DPP])[1]


def test_built_library_is_hazard_free():
    import __graft_entry__ as g
    g.build_hip()
    assert chk.main(g.HIP_LIB) == 0
