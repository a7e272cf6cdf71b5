"""Prototype (numpy) of the CONDENSED Newton solve the 16-lane kernel runs on models with potentiometer rows
(DESIGN.md "Condensing the linear rows"): a measuring instrument and the executable statement of the algebra,
not part of the product.  Compares, sample by sample, a run of HomotopySolver{SimpleSolver} in condensed form
against the oracle (same stack): outputs and Newton iteration totals.

Residual rows of potentiometers, res = v - r w i with w = pos or 1 - pos (src/elements.jl:25-30), are LINEAR in z
for a given p whenever the fq row of `pos` is zero (pos is driven by an input only).  With static pivot columns
Lc (one per linear row) and z = (z_L, z_N):

    A_L(pos) z + b_L(p) = 0,  A_L = fq[v] - r w fq[i],  b_L = pfull[v] - r w pfull[i]
    z_L = zp_L - W z_N,       W = A_LL^-1 A_LN,         zp_L = -A_LL^-1 b_L
    q   = pfull + fq z = pf' + fq' z_N,   pf' = pfull + fq[:, Lc] zp_L,   fq' = fq[:, Nc] - fq[:, Lc] W

Newton on the nn - nl nonlinear rows in z_N is the reference's Newton on the full system (src/solvers.jl:207-236)
restricted to iterates that satisfy the linear rows -- which all of the reference's iterates do after their first
step, and the extrapolated start does as long as the pots have not moved since the origin was taken.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def elem_eval(kind, par, q, nn, nq):
    from oracle import refpy
    L = refpy.lib()
    r = np.zeros(4)
    J = np.zeros(32)
    par = np.ascontiguousarray(par, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    L.acme_ref_eval_element(kind, par.ctypes.data_as(dp), q.ctypes.data_as(dp), r.ctypes.data_as(dp), J.ctypes.data_as(dp))
    return r[:nn].copy(), J[:nn * nq].reshape(nn, nq).copy()


def linear_rows(sub):
    """[(row, v, i, pos, r, w0, w1)] for the potentiometer rows whose pos entry does not depend on z."""
    rows = []
    for e in sub.table:
        if e["kind"] != 3:
            continue
        q0 = e["qoff"]
        if np.any(sub.fq[q0 + 4] != 0):
            continue
        for er in range(2):
            rows.append((e["roff"] + er, q0 + er, q0 + 2 + er, q0 + 4, e["par"][0], 0.0 if er == 0 else 1.0,
                         1.0 if er == 0 else -1.0))
    return rows


def choose_columns(sub, rows, pos=0.5):
    """Static pivot columns: Gaussian elimination of A_L(pos) with complete pivoting; returns (row order, columns)."""
    A = np.array([sub.fq[v] - r * (w0 + w1 * pos) * sub.fq[i] for (_, v, i, _, r, w0, w1) in rows])
    # equilibrate rows so that the pivot search compares like with like
    A = A / np.abs(A).max(axis=1, keepdims=True)
    nl, nn = A.shape
    rleft, cols, order = list(range(nl)), [], []
    for _ in range(nl):
        sub_ = np.abs(A[rleft])
        k = np.unravel_index(np.argmax(sub_), sub_.shape)
        pr, pc = rleft[k[0]], int(k[1])
        order.append(pr)
        cols.append(pc)
        rleft.remove(pr)
        for r_ in rleft:
            A[r_] -= A[r_, pc] / A[pr, pc] * A[pr]
        A[:, pc] = 0.0
    return order, cols


class Condensed:
    def __init__(self, model, exact=True):
        self.m = model
        self.s = s = model.subs[0]
        self.rows = linear_rows(s)
        self.order, self.Lc = choose_columns(s, self.rows)
        self.Lr = [r[0] for r in self.rows]
        self.Nr = [r for r in range(s.nn) if r not in self.Lr]
        self.Nc = [c for c in range(s.nn) if c not in self.Lc]
        self.posval = None
        self.exact = exact
        self.tol, self.maxiter = 1e-10, 500
        # origin
        self.lp = np.zeros(s.np)
        self.lz = s.init_z.copy()
        self.origin = None
        self.iters = 0
        self.n_recond = 0

    # --- element rows --------------------------------------------------------------------
    def eval_all(self, q):
        s = self.s
        res = np.zeros(s.nn)
        Jq = np.zeros((s.nn, s.nq))
        for e in s.table:
            r, J = elem_eval(e["kind"], e["par"], q[e["qoff"]:e["qoff"] + e["nq"]], e["nn"], e["nq"])
            res[e["roff"]:e["roff"] + e["nn"]] = r
            Jq[e["roff"]:e["roff"] + e["nn"], e["qoff"]:e["qoff"] + e["nq"]] = J
        return res, Jq

    # --- condensation ----------------------------------------------------------------------
    def condense(self, pf):
        s = self.s
        pos = np.array([pf[r[3]] for r in self.rows])
        if self.posval is not None and np.array_equal(pos, self.posval):
            return False
        self.posval = pos
        self.n_recond += 1
        self.Rw = np.array([r * (w0 + w1 * pf[p]) for (_, v, i, p, r, w0, w1) in self.rows])
        self.AL = np.array([s.fq[v] - rw * s.fq[i] for (_, v, i, _, _, _, _), rw in zip(self.rows, self.Rw)])
        self.ALL = self.AL[:, self.Lc]
        self.W = np.linalg.solve(self.ALL, self.AL[:, self.Nc])
        self.fqr = s.fq[:, self.Nc] - s.fq[:, self.Lc] @ self.W
        return True

    def bL(self, pf):
        return np.array([pf[v] - rw * pf[i] for (_, v, i, _, _, _, _), rw in zip(self.rows, self.Rw)])

    def full_z(self, zN, zpL):
        z = np.zeros(self.s.nn)
        z[self.Nc] = zN
        z[self.Lc] = zpL - self.W @ zN
        return z

    # --- solve(::SimpleSolver, p) in condensed form ----------------------------------------
    def set_origin(self, p, z):
        """set_extrapolation_origin(solver, p, z): linearise at (p, z) (z taken as it is)."""
        s = self.s
        pf = s.q0 + s.pexp @ p
        self.condense(pf)
        res, Jq = self.eval_all(pf + s.fq @ z)
        self.origin = dict(pos=self.posval.copy(), Jq=Jq, pf=pf, J=Jq @ s.fq)
        self.lp, self.lz = p.copy(), z.copy()

    def simple_solve(self, p):
        s = self.s
        if self.origin is None:
            self.set_origin(self.lp, self.lz)
        pf = s.q0 + s.pexp @ p
        o = self.origin
        # extrapolated start, the reference's (full form; the kernel's reduced replay is algebraically this when
        # the pots have not moved since the origin)
        z0 = self.lz - np.linalg.solve(o["J"], o["Jq"] @ (pf - o["pf"]))
        moved = self.condense(pf)
        zpL = -np.linalg.solve(self.ALL, self.bL(pf))
        pfr = pf + s.fq[:, self.Lc] @ zpL
        zN = z0[self.Nc].copy()
        z = z0
        offsub = True      # is z (possibly) off the subspace of the linear rows?  First iterate: decided below
        if not self.exact:
            z = self.full_z(zN, zpL)
            offsub = False
        conv = False
        it = 0
        for it in range(1, self.maxiter + 1):
            if offsub:
                q = pf + s.fq @ z
            else:
                q = pfr + self.fqr @ zN
            res, Jq = self.eval_all(q)
            resN = res[self.Nr]
            if offsub:
                resmax = np.abs(res).max()
            else:
                resmax = np.abs(resN).max()
            if not np.isfinite(resmax):
                break
            S = Jq[self.Nr] @ self.fqr
            if not np.all(np.isfinite(S)):
                break
            if resmax < self.tol:
                conv = True
                break
            rhs = resN
            if offsub:
                # z1 = zp + N w1,  S w1 = J_N (z0 - zp) - F_N   <=>   S dz_N = F_N - Jq_N (q - q'),  q' = pf' + fq' z_N
                qproj = pfr + self.fqr @ zN
                rhs = resN - Jq[self.Nr] @ (q - qproj)
            try:
                dz = np.linalg.solve(S, rhs)
            except np.linalg.LinAlgError:
                break
            zN = zN - dz
            z = self.full_z(zN, zpL)
            offsub = False
        self.iters = it
        self.z = z
        if conv:
            res, Jq = self.eval_all(pf + s.fq @ z)
            self.origin = dict(pos=self.posval.copy(), Jq=Jq, pf=pf, J=Jq @ s.fq)
            self.lp, self.lz = p.copy(), z.copy()
        return conv

    def homotopy_solve(self, p):
        conv = self.simple_solve(p)
        its = self.iters
        if not conv:
            a, best = 0.5, 0.0
            start = self.lp.copy()
            while best < 1:
                pa = start * (1 - a) + a * p
                conv = self.simple_solve(pa)
                its += self.iters
                if conv:
                    best, a = a, 1.0
                else:
                    na = (a + best) / 2
                    if not (best < na < a):
                        break
                    a = na
        self.h_iters = its
        return conv

    def run(self, u):
        m, s = self.m, self.s
        T = u.shape[1]
        y = np.zeros((m.ny, T))
        x = np.zeros(m.nx)
        total, warn = 0, 0
        for n in range(T):
            p = s.dq @ x + s.eq @ u[:, n]
            conv = self.homotopy_solve(p)
            total += self.h_iters
            warn += 0 if conv else 1
            z = self.z
            y[:, n] = m.y0 + m.dy @ x + m.ey @ u[:, n] + m.fy @ z
            x = m.x0 + m.a @ x + m.b @ u[:, n] + m.c @ z
        return y, total, warn


def main():
    from helpers import HS, load, sine, sweep_inputs
    from oracle.refpy import RefRunner
    m = load("superover_var", HS)
    c0 = Condensed(m)
    print("linear rows", c0.Lr, "order", c0.order, "pivot columns", c0.Lc)
    T = int(os.environ.get("T", "300"))
    cases = {"steady": sweep_inputs("superover_var", 3, T, seed=1),
             "ramps": np.stack([sine(T), np.linspace(0.97, 0, T), np.linspace(0, 1, T), np.linspace(1, 0, T)])[None]}
    for name, U in cases.items():
        for i in range(U.shape[0]):
            ref = RefRunner(m)
            yr = ref.run(U[i])
            for exact in (True, False):
                c = Condensed(m, exact=exact)
                y, tot, warn = c.run(U[i])
                print(f"{name}[{i}] exact={exact}: max|dy| {np.abs(y - yr).max():.3e}  iterations {tot} vs oracle "
                      f"{ref.report.iters_total}  warn {warn}/{ref.report.n_warn}  recondensations {c.n_recond}")


if __name__ == "__main__":
    main()
