"""Construction-time nonlinear solve on the host (pure Python floats).

The reference's model constructor needs one Newton/homotopy solve per nonlinear
sub-problem to find the initial operating point (``initial_solution``,
src/ACME.jl:453-464) and ``steadystate`` needs the same machinery
(src/ACME.jl:474-497).  These are one-off, per-model computations that belong to the
derivation front end, not to the per-sample hot path, so they run here on the host.
The per-sample path (``run!``) never calls into this module: it runs on the GPU only.

Follows: LinearSolver src/solvers.jl:38-132, SimpleSolver :151-236,
HomotopySolver :247-302, element functions src/elements.jl (lines cited per kind).
"""
from __future__ import annotations

import math

from .circuit import (KIND_BJT, KIND_DIODE, KIND_JA, KIND_MACAK, KIND_MOSFET, KIND_POT)

INF = math.inf


def _exp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return INF


def _evalpoly(x, coeffs):
    acc = 0.0
    for c in reversed(coeffs):
        acc = acc * x + c
    return acc


def eval_element(kind, par, q):
    """Return (res, Jq) for one element; res: list[nn], Jq: list[nn][nq]."""
    if kind == KIND_DIODE:                                   # src/elements.jl:238-244
        is_, eta = par[0], par[1]
        v, i = q
        ex = _exp(v * (1 / (25e-3 * eta)))
        return [is_ * (ex - 1) - i], [[is_ / (25e-3 * eta) * ex, -1.0]]
    if kind == KIND_POT:                                     # src/elements.jl:25-30
        r = par[0]
        v1, v2, i1, i2, pos = q
        return ([v1 - r * pos * i1, v2 - r * (1 - pos) * i2],
                [[1.0, 0.0, -r * pos, 0.0, -r * i1],
                 [0.0, 1.0, 0.0, -r * (1 - pos), -r * i2]])
    if kind == KIND_BJT:                                     # src/elements.jl:323-401
        (ise, isc, etae, etac, bf, br, ile, ilc, etael, etacl, vaf, var, ikf, ikr) = par[:14]
        vE, vC, iE, iC = q
        expE = _exp(vE * (1 / (25e-3 * etae)))
        expC = _exp(vC * (1 / (25e-3 * etac)))
        i_f = (bf / (1 + bf) * ise) * (expE - 1)
        i_r = (br / (1 + br) * isc) * (expC - 1)
        di_f1 = (bf / (1 + bf) * ise / (25e-3 * etae)) * expE
        di_r2 = (br / (1 + br) * isc / (25e-3 * etac)) * expC
        early = not (var == INF and vaf == INF)
        knee = not (ikf == INF and ikr == INF)
        if not early and not knee:
            i_cc = i_f - i_r
            di_cc1 = di_f1
            di_cc2 = -di_r2
        elif early and not knee:
            q1i = 1 - vE * (1 / var) - vC * (1 / vaf)
            i_cc = q1i * (i_f - i_r)
            di_cc1 = (-1 / var) * (i_f - i_r) + q1i * di_f1
            di_cc2 = (-1 / vaf) * (i_f - i_r) - q1i * di_r2
        elif not early and knee:
            q2 = i_f * (1 / ikf) + i_r * (1 / ikr)
            qden = 1 + math.sqrt(1 + 4 * q2)
            qfact = 2 / qden
            i_cc = qfact * (i_f - i_r)
            dq21 = di_f1 * (1 / ikf)
            dq22 = di_r2 * (1 / ikr)
            dqfact1 = -4 * dq21 / (qden - 1) / (qden ** 2)
            dqfact2 = -4 * dq22 / (qden - 1) / (qden ** 2)
            di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1
            di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2
        else:
            q1i = 1 - vE * (1 / var) - vC * (1 / vaf)
            q2 = i_f * (1 / ikf) + i_r * (1 / ikr)
            qden = 1 + math.sqrt(1 + 4 * q2)
            qfact = 2 * q1i / qden
            i_cc = qfact * (i_f - i_r)
            dq1i1 = -1 / var
            dq1i2 = -1 / vaf
            dq21 = di_f1 * (1 / ikf)
            dq22 = di_r2 * (1 / ikr)
            dqfact1 = (2 * dq1i1 * qden - q1i * 4 * dq21 / (qden - 1)) / (qden ** 2)
            dqfact2 = (2 * dq1i2 * qden - q1i * 4 * dq22 / (qden - 1)) / (qden ** 2)
            di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1
            di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2
        iBE = (1 / bf) * i_f
        diBE1 = (1 / bf) * di_f1
        if ile != 0:
            expEl = _exp(vE * (1 / (25e-3 * etael))) if etael != etae else expE
            iBE += ile * (expEl - 1)
            diBE1 += (ile / (25e-3 * etae)) * expEl
        iBC = (1 / br) * i_r
        diBC2 = (1 / br) * di_r2
        if ilc != 0:
            expCl = _exp(vC * (1 / (25e-3 * etacl))) if etacl != etac else expC
            iBC += ilc * (expCl - 1)
            diBC2 += (ilc / (25e-3 * etac)) * expCl
        return ([i_cc + iBE - iE, -i_cc + iBC - iC],
                [[di_cc1 + diBE1, di_cc2, -1.0, 0.0],
                 [-di_cc1, -di_cc2 + diBC2, 0.0, -1.0]])
    if kind == KIND_MOSFET:                                  # src/elements.jl:453-479
        pol, lam = par[0], par[1]
        nvt = int(par[2]); vt = par[3:3 + nvt]
        na = int(par[7]); al = par[8:8 + na]
        dvt = [vt[k] * k for k in range(1, nvt)]
        dal = [al[k] * k for k in range(1, na)]
        vgs, vds, id_ = q
        a_ = _evalpoly(pol * vgs, al)
        da = _evalpoly(pol * vgs, dal) if dal else 0.0
        vt_ = _evalpoly(pol * vgs, vt)
        dvt_ = _evalpoly(pol * vgs, dvt) if dvt else 0.0
        lam_ = lam if vds >= 0 else 0.0
        if vgs <= vt_:
            return [-id_], [[0.0, 0.0, -1.0]]
        if vds <= vgs - vt_:
            return ([a_ * (vgs - vt_ - 0.5 * vds) * vds * (1 + lam_ * vds) - id_],
                    [[a_ * (1 - dvt_) * vds * (1 + lam_ * vds)
                      + da * (vgs - vt_ - 0.5 * vds) * vds * (1 + lam_ * vds),
                      a_ * (vgs - vt_ + vds * (2 * lam_ * (vgs - vt_ - 0.75 * vds) - 1)),
                      -1.0]])
        return ([(a_ / 2) * (vgs - vt_) ** 2 * (1 + lam_ * vds) - id_],
                [[a_ * (vgs - vt_) * (1 - dvt_) * (1 + lam_ * vds)
                  + da / 2 * (vgs - vt_) ** 2 * (1 + lam_ * vds),
                  lam_ * a_ / 2 * (vgs - vt_) ** 2, -1.0]])
    if kind == KIND_MACAK:                                   # src/elements.jl:540-546
        gain, scale = par[0], par[1]
        vi, vo = q
        vs = vi * (gain / scale)
        ch = math.cosh(vs) if abs(vs) < 700 else INF
        return [math.tanh(vs) * scale - vo], [[gain / ch ** 2 if ch != INF else 0.0, -1.0]]
    if kind == KIND_JA:                                      # src/elements.jl:107-129
        Ms, a, alpha, c, k = par[:5]
        q1, q2, q3, q4 = q
        coth = (1 / math.tanh(q1)) if q1 != 0 else INF
        aq1 = abs(q1)
        L = q1 / 3 if aq1 < 1e-4 else coth - 1 / q1
        Ld = 1 / 3 if aq1 < 1e-4 else 1 / q1 ** 2 - coth ** 2 + 1
        Ld2 = -2 / 15 * q1 if aq1 < 1e-3 else 2 * coth * (coth ** 2 - 1) - 2 / q1 ** 3
        delta = 1.0 if q3 > 0 else -1.0
        Man = Ms * L
        sgn = lambda v: (v > 0) - (v < 0)
        dM = 1.0 if sgn(q3) == sgn(Man - q2) else 0.0
        den = delta * (k * (1 - c)) - alpha * (Man - q2)
        res = [(1e-4 / Ms) * ((1 - c) * dM * (Man - q2) / den * q3
                              + (c * Ms / a) * (q3 + alpha * q4) * Ld - q4)]
        J11 = (1e-4 / Ms) * (((1 - c) ** 2 * k * Ms) * dM * Ld * delta / den ** 2 * q3
                              + (c * Ms / a) * (q3 + alpha * q4) * Ld2)
        J12 = (1e-4 / Ms) * -(1 - c) ** 2 * k * dM * delta / den ** 2 * q3
        J13 = (1e-4 / Ms) * ((1 - c) * dM * (Man - q2) / den + (c * Ms / a) * Ld)
        J14 = (1e-4 / Ms) * ((c * Ms / a * alpha) * Ld - 1)
        return res, [[J11, J12, J13, J14]]
    raise ValueError(f"unknown element kind {kind}")


def eval_table(table, q, nn, nq):
    """CircuitNLFunc (src/circuit.jl:6-20): residual vcat + block-diagonal Jq."""
    res = [0.0] * nn
    Jq = [[0.0] * nq for _ in range(nn)]
    for e in table:
        r, J = eval_element(e["kind"], e["par"], q[e["qoff"]:e["qoff"] + e["nq"]])
        for i in range(e["nn"]):
            res[e["roff"] + i] = r[i]
            for j in range(e["nq"]):
                Jq[e["roff"] + i][e["qoff"] + j] = J[i][j]
    return res, Jq


# --- LinearSolver (src/solvers.jl:38-132) -------------------------------------------
def lu_factor(A):
    n = len(A)
    f = [list(r) for r in A]
    ipiv = [0] * n
    for k in range(n):
        kp = k
        amax = 0.0
        for i in range(k, n):
            ab = abs(f[i][k])
            if ab > amax:
                kp, amax = i, ab
        ipiv[k] = kp
        if f[kp][k] != 0.0:
            if k != kp:
                f[k], f[kp] = f[kp], f[k]
            inv = 1.0 / f[k][k]
            f[k][k] = inv
            for i in range(k + 1, n):
                f[i][k] *= inv
        else:
            return None
        for j in range(k + 1, n):
            fkj = f[k][j]
            for i in range(k + 1, n):
                f[i][j] -= f[i][k] * fkj
    return f, ipiv


def lu_solve(lu, b):
    f, ipiv = lu
    n = len(f)
    x = list(b)
    for i in range(n):
        x[i], x[ipiv[i]] = x[ipiv[i]], x[i]
    for j in range(n):
        xj = x[j]
        for i in range(j + 1, n):
            x[i] -= f[i][j] * xj
    for j in range(n - 1, -1, -1):
        xj = x[j] = f[j][j] * x[j]
        for i in range(j):
            x[i] -= f[i][j] * xj
    return x


class HostNleq:
    """ParametricNonLinEq bound to (q0, pexp, fq, element table); q = q0 + pexp*p + fq*z."""

    def __init__(self, table, fq, q0=None, pexp=None):
        self.table = table
        self.fq = fq                       # nq x nn floats
        self.nq = len(fq)
        self.nn = len(fq[0]) if self.nq else sum(e["nn"] for e in table)
        if pexp is None:                   # 3-arg ParametricNonLinEq: p == q (src/solvers.jl:23-28)
            self.pexp = None
            self.np = self.nq
            self.q0 = [0.0] * self.nq
        else:
            self.pexp = pexp
            self.np = len(pexp[0]) if self.nq else 0
            self.q0 = list(q0)

    def set_p(self, p):
        if self.pexp is None:
            self.pfull = list(p)
        else:
            self.pfull = [q0 + sum(a * b for a, b in zip(row, p))
                          for q0, row in zip(self.q0, self.pexp)]

    def evaluate(self, z):
        q = [pf + sum(a * b for a, b in zip(row, z)) for pf, row in zip(self.pfull, self.fq)]
        res, Jq = eval_table(self.table, q, self.nn, self.nq)
        self.Jq = Jq
        J = [[sum(Jq[i][c] * self.fq[c][j] for c in range(self.nq) if Jq[i][c] != 0.0)
              for j in range(self.nn)] for i in range(self.nn)]
        return res, J

    def calc_Jp(self):
        if self.pexp is None:
            return [list(r) for r in self.Jq]
        return [[sum(self.Jq[i][c] * self.pexp[c][j] for c in range(self.nq) if self.Jq[i][c] != 0.0)
                 for j in range(self.np)] for i in range(self.nn)]


class HostSimpleSolver:
    """SimpleSolver (src/solvers.jl:151-236)."""

    def __init__(self, nleq, initial_p, initial_z, tol=1e-10):
        self.nleq = nleq
        self.tol = tol
        self.iters = 0
        self.resmaxabs = 0.0
        self.z = list(initial_z)
        self.set_extrapolation_origin(initial_p, initial_z)

    def set_extrapolation_origin(self, p, z):
        self.nleq.set_p(p)
        res, J = self.nleq.evaluate(z)
        self.J = J
        self.last_lu = lu_factor(J)
        self.last_Jp = self.nleq.calc_Jp()
        self.last_p = list(p)
        self.last_z = list(z)

    def hasconverged(self):
        return self.resmaxabs < self.tol

    def solve(self, p, maxiter=500):
        nl = self.nleq
        nl.set_p(p)
        dp = [a - b for a, b in zip(p, self.last_p)]
        t = [sum(a * b for a, b in zip(row, dp)) for row in self.last_Jp]
        if self.last_lu is not None and nl.nn:
            t = lu_solve(self.last_lu, t)
        z = [a - b for a, b in zip(self.last_z, t)]
        lu = None
        self.iters = 0
        for it in range(1, maxiter + 1):
            self.iters = it
            res, J = nl.evaluate(z)
            self.J = J
            self.resmaxabs = max((abs(r) for r in res), default=0.0)
            if not math.isfinite(self.resmaxabs) or \
                    not all(math.isfinite(v) for row in J for v in row):
                self.z = z
                return z
            lu = lu_factor(J)
            if lu is None:
                self.z = z
                return z
            if self.hasconverged():
                break
            d = lu_solve(lu, res)
            z = [a - b for a, b in zip(z, d)]
        if self.hasconverged():
            self.last_Jp = nl.calc_Jp()
            self.last_lu = lu
            self.last_p = list(p)
            self.last_z = list(z)
        self.z = z
        return z


class HostHomotopySolver:
    """HomotopySolver{SimpleSolver} (src/solvers.jl:247-302)."""

    def __init__(self, nleq, initial_p, initial_z, tol=1e-10):
        self.base = HostSimpleSolver(nleq, initial_p, initial_z, tol)
        self.iters = 0

    def hasconverged(self):
        return self.base.hasconverged()

    def solve(self, p):
        b = self.base
        z = b.solve(p)
        self.iters = b.iters
        if not b.hasconverged():
            a = 0.5
            best_a = 0.0
            start_p = list(b.last_p)
            while best_a < 1:
                pa = [(1 - a) * s + a * t for s, t in zip(start_p, p)]
                # reference: pa = start_p*(1-a); pa += a*p  (same two roundings per entry)
                z = b.solve(pa)
                self.iters += b.iters
                if b.hasconverged():
                    best_a = a
                    a = 1.0
                else:
                    new_a = (a + best_a) / 2
                    if not (best_a < new_a < a):
                        break
                    a = new_a
        return z


def initial_solution(table, fq, q0):
    """src/ACME.jl:453-464: homotopy on q from 0 to q0, z starting at 0."""
    nleq = HostNleq(table, fq)
    solver = HostHomotopySolver(nleq, [0.0] * nleq.nq, [0.0] * nleq.nn)
    z = solver.solve(list(q0))
    if not solver.hasconverged():
        raise RuntimeError("Failed to find initial solution")
    return z
