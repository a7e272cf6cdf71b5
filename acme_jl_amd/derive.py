"""Circuit -> discrete-time nonlinear state-space matrices (exact rational, host side).

Restates the reference's derivation so that the GPU hot path has inputs to run on:
  * model_matrices ............... src/ACME.jl:264-315
  * tryextract / nldecompose! .... src/ACME.jl:319-378
  * split_nl_model_matrices ...... src/ACME.jl:381-401
  * reduce_pdims! ................ src/ACME.jl:403-451
  * DiscreteModel constructor .... src/ACME.jl:150-262 (initial solutions, folding of
    constant sub-problems, Rational -> Float64 conversion)

The *basis* chosen for z and p depends on net/row ordering details (Julia Dict hash
order for pins) that cannot be reproduced; outputs y and all dimensions (np, nn pins in
test/runtests.jl) are basis independent.
"""
from __future__ import annotations

import itertools
import warnings
from fractions import Fraction

from . import ratmat as rm
from .hostsolve import initial_solution


def consecranges(lengths):
    out = []
    e = 0
    for l in lengths:
        out.append(list(range(e, e + l)))
        e += l
    return out


def model_matrices(circ, t):
    """src/ACME.jl:264-315.  Returns a dict of dense Fraction matrices/vectors."""
    t = Fraction(t)
    nb, nx, nq, nu = circ.nb, circ.nx, circ.nq, circ.nu
    mv, mi, mx, mxd, mq, mu = (circ.blockdiag(k) for k in ("mv", "mi", "mx", "mxd", "mq", "mu"))
    u0 = circ.u0()
    nl = len(mv)
    tv, ti = circ.topomat()
    tv = rm.frac_matrix(tv)
    ti = rm.frac_matrix(ti)
    half = Fraction(1, 2)
    mxx = rm.add(rm.scale(mxd, 1 / t), rm.scale(mx, half)) if nx else [[] for _ in range(nl)]
    mxr = rm.sub(rm.scale(mxd, 1 / t), rm.scale(mx, half)) if nx else [[] for _ in range(nl)]
    top = rm.hcat(mv, mi, mxx, mq) if nl else []
    # blockdiag(tv, ti) followed by zero columns for x and q
    bd = []
    for row in tv:
        bd.append(list(row) + [rm.ZERO] * nb + [rm.ZERO] * (nx + nq))
    for row in ti:
        bd.append([rm.ZERO] * nb + list(row) + [rm.ZERO] * (nx + nq))
    lhs = rm.vcat(top, bd)
    rhs_top = rm.hcat(u0, mu, mxr) if nl else []
    rhs = rm.vcat(rhs_top, rm.zeros(nb, 1 + nu + nx))
    ncols = 2 * nb + nx + nq
    x, f = rm.gensolve(lhs, rhs, ncols, 1 + nu + nx)

    rowranges = consecranges((nb, nb, nx, nq))
    fq = rm.rows(f, rowranges[3])
    kf = len(f[0]) if f else 0
    ns = rm.nullspace(fq, kf) if kf else []
    nns = len(ns[0]) if ns else 0
    n_indet = nns
    if nns:
        indet = rm.matmul(f, ns)
        if sum(float(v) ** 2 for r in rowranges[2] for v in indet[r]) > 1e-20:
            warnings.warn("State update depends on indeterminate quantity")
    else:
        indet = rm.zeros(ncols, 0)
    while nns > 0:
        i, j = rm.argmax_abs(ns)
        ns = [[v for cc, v in enumerate(row) if cc != j] for r, row in enumerate(ns) if r != i]
        f = [[v for cc, v in enumerate(row) if cc != i] for row in f]
        nns -= 1
    kf = len(f[0]) if f else 0

    fv, fi, c, fq = (rm.rows(f, r) for r in rowranges)
    colranges = consecranges((1, nu, nx))
    blocks = {}
    names = (("v0", "i0", "x0", "q0"), ("ev", "ei", "b", "eq_full"), ("dv", "di", "a", "dq_full"))
    for cr, nms in zip(colranges, names):
        for rr, nm in zip(rowranges, nms):
            blocks[nm] = rm.sub_block(x, rr, cr)
    for v in ("v0", "i0", "x0", "q0"):
        blocks[v] = [row[0] for row in blocks[v]]

    pv, pi_, px, pxd, pq = (circ.blockdiag(k) for k in ("pv", "pi", "px", "pxd", "pq"))
    ny = len(pv)
    pxx = rm.add(rm.scale(px, half), rm.scale(pxd, 1 / t)) if nx else [[] for _ in range(ny)]
    p = rm.hcat(pv, pi_, pxx, pq) if ny else []
    if ny and n_indet:
        pind = rm.matmul(p, indet)
        if sum(float(v) ** 2 for row in pind for v in row) > 1e-20:
            warnings.warn("Model output depends on indeterminate quantity")
    xd = rm.cols(x, colranges[2])
    dy = rm.matmul(p, xd, nx)
    if nx and ny:
        dy = rm.add(dy, rm.sub(rm.scale(px, half), rm.scale(pxd, 1 / t)))
    ey = rm.matmul(p, rm.cols(x, colranges[1]), nu)
    fy = rm.matmul(p, f, kf)
    y0 = [row[0] for row in rm.matmul(p, rm.cols(x, colranges[0]), 1)]
    mats = dict(fv=fv, fi=fi, c=c, fq=fq, dy=dy, ey=ey, fy=fy, y0=y0,
                nx=nx, nu=nu, ny=ny, nz=kf)
    mats.update(blocks)
    return mats


def tryextract(fq, numcols):
    """src/ACME.jl:319-347.  ``fq`` is a private copy (rows x ncols)."""
    ncols = len(fq[0]) if fq else 0
    a = rm.eye(ncols)
    if numcols >= ncols:
        return a
    fq = [list(r) for r in fq]
    for colcnt in range(numcols):
        if not fq:
            return None
        sub = [row[colcnt:] for row in fq]
        i, j = rm.argmax_abs(sub)
        j += colcnt
        piv = fq[i][j]
        if not piv:
            return None
        for row in fq:
            row[colcnt], row[j] = row[j], row[colcnt]
        for row in a:
            row[colcnt], row[j] = row[j], row[colcnt]
        piv = fq[i][colcnt]
        fac = [fq[i][jj] / piv for jj in range(colcnt + 1, ncols)]
        for row in a:
            ac = row[colcnt]
            if ac:
                for k, jj in enumerate(range(colcnt + 1, ncols)):
                    if fac[k]:
                        row[jj] -= ac * fac[k]
        for row in fq:
            fc = row[colcnt]
            if fc:
                for k, jj in enumerate(range(colcnt + 1, ncols)):
                    if fac[k]:
                        row[jj] -= fc * fac[k]
        del fq[i]
        if all(not v for row in fq for v in row[colcnt + 1:]):
            return a
    return None


def nldecompose(mats, nns, nqs):
    """src/ACME.jl:349-378; updates mats['fq'], mats['c'], mats['fy'] in place."""
    fq = mats["fq"]
    ncols = mats["nz"]
    a = rm.eye(ncols)
    sub_ranges = consecranges(nqs)
    extracted = []
    rem_first = 0
    rem_nles = sorted(e for e in range(len(nqs)) if nqs[e] > 0)
    while rem_nles:
        found = False
        for sz in range(1, len(rem_nles) + 1):
            for sub in itertools.combinations(rem_nles, sz):
                nn_sub = sum(nns[e] for e in sub)
                ridx = [r for e in sub for r in sub_ranges[e]]
                rem_cols = list(range(rem_first, ncols))
                a_update = tryextract(rm.sub_block(fq, ridx, rem_cols), nn_sub)
                if a_update is not None:
                    newfq = rm.matmul(rm.cols(fq, rem_cols), a_update, len(rem_cols))
                    for r, row in enumerate(fq):
                        row[rem_first:] = newfq[r]
                    newa = rm.matmul(rm.cols(a, rem_cols), a_update, len(rem_cols))
                    for r, row in enumerate(a):
                        row[rem_first:] = newa[r]
                    rem_first += nn_sub
                    extracted.append(list(sub))
                    rem_nles = [e for e in rem_nles if e not in sub]
                    found = True
                    break
            if found:
                break
        if not found:
            raise RuntimeError("nonlinearity decomposition failed")
    mats["c"] = rm.matmul(mats["c"], a, ncols)
    mats["fy"] = rm.matmul(mats["fy"], a, ncols)
    return extracted


def split_nl_model_matrices(mats, model_qidxs, model_nns):
    """src/ACME.jl:381-401"""
    zr = consecranges(model_nns)
    nsub = len(model_qidxs)
    fq = mats["fq"]
    fqs, fqprev_fulls = [], []
    for i, qidxs in enumerate(model_qidxs):
        fqs.append(rm.sub_block(fq, qidxs, zr[i]))
        prev = [c for k in range(i) for c in zr[k]]
        rest = sum(model_nns[i:])
        fqprev_fulls.append([[fq[r][c] for c in prev] + [rm.ZERO] * rest for r in qidxs])
    return dict(
        dq_fulls=[rm.rows(mats["dq_full"], q) for q in model_qidxs],
        eq_fulls=[rm.rows(mats["eq_full"], q) for q in model_qidxs],
        fqs=fqs, fqprev_fulls=fqprev_fulls,
        q0s=[[mats["q0"][r] for r in q] for q in model_qidxs],
        nsub=nsub)


def reduce_pdims(mats):
    """src/ACME.jl:403-451"""
    subcount = len(mats["dq_fulls"])
    nx, nu = mats["nx"], mats["nu"]
    nzt = len(mats["fqprev_fulls"][0][0]) if subcount and mats["fqprev_fulls"][0] else \
        sum(len(f[0]) if f else 0 for f in mats["fqs"])
    dqs, eqs, fqprevs, pexps = [None] * subcount, [None] * subcount, [None] * subcount, [None] * subcount
    offset = 0
    for idx in range(subcount):
        nqi = len(mats["dq_fulls"][idx])
        nzp = len(mats["fqprev_fulls"][idx][0]) if nqi else nzt
        big = [list(mats["dq_fulls"][idx][r]) + list(mats["eq_fulls"][idx][r])
               + list(mats["fqprev_fulls"][idx][r]) for r in range(nqi)]
        wid = nx + nu + nzp
        pexp, dqeq = rm.rank_factorize(big, wid)
        pexps[idx] = pexp
        dqs[idx] = [row[:nx] for row in dqeq]
        eqs[idx] = [row[nx:nx + nu] for row in dqeq]
        fqprevs[idx] = [row[nx + nu:] for row in dqeq]

        fq = mats["fqs"][idx]
        nn = len(fq[0]) if fq else 0
        fqt = rm.transpose(fq, nn)
        fq_pinv = rm.gensolve(rm.matmul(fqt, fq, nn), fqt, nn, nqi)[0]     # nn x nq
        np_old = len(pexp[0]) if pexp else 0
        proj = rm.sub(pexp, rm.matmul(fq, rm.matmul(fq_pinv, pexp, np_old), np_old)) if nqi else pexp
        pexp2, f = rm.rank_factorize(proj, np_old)
        np_new = len(pexp2[0]) if pexp2 else 0
        if nqi and np_new < np_old:
            cols = list(range(offset, offset + nn))
            pin = rm.matmul(fq_pinv, pexps[idx], np_old)                   # nn x np_old
            ccols = rm.cols(mats["c"], cols)
            fycols = rm.cols(mats["fy"], cols)
            mats["a"] = rm.sub(mats["a"], rm.matmul(ccols, rm.matmul(pin, dqs[idx], nx), nx))
            mats["b"] = rm.sub(mats["b"], rm.matmul(ccols, rm.matmul(pin, eqs[idx], nu), nu))
            mats["dy"] = rm.sub(mats["dy"], rm.matmul(fycols, rm.matmul(pin, dqs[idx], nx), nx))
            mats["ey"] = rm.sub(mats["ey"], rm.matmul(fycols, rm.matmul(pin, eqs[idx], nu), nu))
            for idx2 in range(idx + 1, subcount):
                q = rm.matmul(rm.cols(mats["fqprev_fulls"][idx2], cols), pin, np_old)
                mats["dq_fulls"][idx2] = rm.sub(mats["dq_fulls"][idx2], rm.matmul(q, dqs[idx], nx))
                mats["eq_fulls"][idx2] = rm.sub(mats["eq_fulls"][idx2], rm.matmul(q, eqs[idx], nu))
                if offset:
                    upd = rm.matmul(q, [row[:offset] for row in fqprevs[idx]], offset)
                    for r, row in enumerate(mats["fqprev_fulls"][idx2]):
                        for cc in range(offset):
                            row[cc] -= upd[r][cc]
            pexps[idx] = pexp2
            dqs[idx] = rm.matmul(f, dqs[idx], nx)
            eqs[idx] = rm.matmul(f, eqs[idx], nu)
            fqprevs[idx] = rm.matmul(f, fqprevs[idx], nzp)
            mats["dq_fulls"][idx] = rm.matmul(pexp2, dqs[idx], nx)
            mats["eq_fulls"][idx] = rm.matmul(pexp2, eqs[idx], nu)
            mats["fqprev_fulls"][idx] = rm.matmul(pexp2, fqprevs[idx], nzp)
        offset += nn
    mats["dqs"], mats["eqs"], mats["fqprevs"], mats["pexps"] = dqs, eqs, fqprevs, pexps
    return mats


def derive(circ, t, decompose_nonlinearity=True):
    """The matrix half of ``DiscreteModel(circ, t, Solver)`` (src/ACME.jl:150-262).

    Returns a plain dict with Float64 (python float) matrices as nested row lists:
      a b c x0 dy ey fy y0, and per sub-problem lists pexps dqs eqs fqprevs fqs q0s,
      tables (element descriptors), init_zs, plus dims.
    """
    mats = model_matrices(circ, t)
    elems = list(circ.elements.values())
    nns = [e.nn for e in elems]
    nqs = [e.nq for e in elems]
    if decompose_nonlinearity:
        nl_elems = nldecompose(mats, nns, nqs)
    else:
        nl_elems = [[i for i, n in enumerate(nns) if n > 0]]
    model_nns = [sum(nns[e] for e in nles) for nles in nl_elems]
    qr = consecranges(nqs)
    model_qidxs = [[r for e in nles for r in qr[e]] for nles in nl_elems]
    mats.update(split_nl_model_matrices(mats, model_qidxs, model_nns))
    mats = reduce_pdims(mats)
    assert circ.nn == sum(model_nns)

    tables = [circ.nonlinear_table(nles) for nles in nl_elems]
    f64 = rm.to_float
    q0s_f = [[float(v) for v in q0] for q0 in mats["q0s"]]
    fqs_f = [f64(m) for m in mats["fqs"]]
    fqprev_f = [f64(m) for m in mats["fqprev_fulls"]]

    init_zs = [[0.0] * n for n in model_nns]
    for idx in range(len(tables)):
        zall = [v for z in init_zs for v in z]
        q = [q0 + sum(a * b for a, b in zip(row, zall)) for q0, row in zip(q0s_f[idx], fqprev_f[idx])]
        init_zs[idx] = initial_solution(tables[idx], fqs_f[idx], q) if model_nns[idx] else []

    while True:
        const_idxs = [i for i, d in enumerate(mats["dqs"]) if len(d) == 0]
        if not const_idxs:
            break
        zr = consecranges(model_nns)
        const_z = [c for i in const_idxs for c in zr[i]]
        varying_z = [c for c in range(sum(model_nns)) if c not in const_z]
        zc = [Fraction(v) for i in const_idxs for v in init_zs[i]]
        for idx in range(len(mats["q0s"])):
            fp = mats["fqprev_fulls"][idx]
            add = rm.matvec(rm.cols(fp, const_z), zc)
            mats["q0s"][idx] = [a + b for a, b in zip(mats["q0s"][idx], add)]
            mats["fqprev_fulls"][idx] = rm.cols(fp, varying_z)
        mats["x0"] = [a + b for a, b in zip(mats["x0"], rm.matvec(rm.cols(mats["c"], const_z), zc))]
        mats["y0"] = [a + b for a, b in zip(mats["y0"], rm.matvec(rm.cols(mats["fy"], const_z), zc))]
        for key in ("q0s", "dq_fulls", "eq_fulls", "fqs", "fqprev_fulls"):
            mats[key] = [m for i, m in enumerate(mats[key]) if i not in const_idxs]
        init_zs = [m for i, m in enumerate(init_zs) if i not in const_idxs]
        model_nns = [m for i, m in enumerate(model_nns) if i not in const_idxs]
        tables = [m for i, m in enumerate(tables) if i not in const_idxs]
        nl_elems = [m for i, m in enumerate(nl_elems) if i not in const_idxs]
        mats["fy"] = rm.cols(mats["fy"], varying_z)
        mats["c"] = rm.cols(mats["c"], varying_z)
        mats = reduce_pdims(mats)

    nsub = len(tables)
    out = dict(
        nx=mats["nx"], nu=mats["nu"], ny=mats["ny"], nsub=nsub,
        nns=list(model_nns),
        nqs=[len(m) for m in mats["pexps"]],
        nps=[len(m) for m in mats["dqs"]],
        a=f64(mats["a"]), b=f64(mats["b"]), c=f64(mats["c"]),
        x0=[float(v) for v in mats["x0"]],
        dy=f64(mats["dy"]), ey=f64(mats["ey"]), fy=f64(mats["fy"]),
        y0=[float(v) for v in mats["y0"]],
        pexps=[f64(m) for m in mats["pexps"]],
        dqs=[f64(m) for m in mats["dqs"]],
        eqs=[f64(m) for m in mats["eqs"]],
        fqprevs=[f64(m) for m in mats["fqprevs"]],
        fqs=[f64(m) for m in mats["fqs"]],
        q0s=[[float(v) for v in q0] for q0 in mats["q0s"]],
        tables=tables, init_zs=init_zs, nl_elems=nl_elems,
    )
    return out
