"""DiscreteModel: the data the hot path runs on (mirror of src/ACME.jl:118-148).

A ``DiscreteModel`` here is plain data: Float64 matrices (numpy, Fortran order so the
memory layout equals Julia's column-major ``Matrix{Float64}``), one element-descriptor
table per nonlinear sub-problem (the C-ABI replacement for the reference's closures,
src/circuit.jl:68-86), each sub-problem's initial extrapolation origin
(p = 0, z = init_z; src/ACME.jl:253-259) and the solver selection (the third positional
argument of the reference constructor, src/ACME.jl:150).

Running a model is not done here: see ``runner.ModelRunner`` (GPU, through the C ABI).
"""
from __future__ import annotations

import json
from fractions import Fraction

import numpy as np

from . import derive as _derive
from .circuit import MAX_ELEM_PAR

# solver selection -- names follow src/solvers.jl
SimpleSolver = "SimpleSolver"
HomotopySolver = "HomotopySolver{SimpleSolver}"
CachingHomotopySolver = "HomotopySolver{CachingSolver{SimpleSolver}}"   # reference default
SOLVER_IDS = {SimpleSolver: 0, HomotopySolver: 1, CachingHomotopySolver: 2}


def _f(m, r, c):
    a = np.zeros((r, c), dtype=np.float64, order="F")
    if r and c:
        a[:, :] = np.asarray(m, dtype=np.float64).reshape(r, c)
    return a


class SubProblem:
    """One nonlinear sub-problem: matrices of src/ACME.jl:123-128 + element table."""

    def __init__(self, nn, nq, np_, pexp, dq, eq, fqprev, fq, q0, init_z, table, row_order=None):
        self.row_order = list(row_order) if row_order else None  # GPU lane assignment hint
        self.nn, self.nq, self.np = nn, nq, np_
        self.pexp, self.dq, self.eq, self.fqprev, self.fq = pexp, dq, eq, fqprev, fq
        self.q0 = np.asarray(q0, dtype=np.float64).reshape(nq)
        self.init_z = np.asarray(init_z, dtype=np.float64).reshape(nn)
        self.table = table  # list of dict(kind, par, nq, nn, qoff, roff)

    def elem_arrays(self):
        n = len(self.table)
        kind = np.array([e["kind"] for e in self.table], dtype=np.int32)
        qoff = np.array([e["qoff"] for e in self.table], dtype=np.int32)
        roff = np.array([e["roff"] for e in self.table], dtype=np.int32)
        par = np.zeros((max(n, 1), MAX_ELEM_PAR), dtype=np.float64)
        for i, e in enumerate(self.table):
            par[i, :len(e["par"])] = e["par"]
        return kind, qoff, roff, par


class DiscreteModel:
    """``DiscreteModel(circ, t, Solver; decompose_nonlinearity=true)`` (src/ACME.jl:150).

    ``solver`` defaults, like the reference's third positional argument, to
    ``HomotopySolver{CachingSolver{SimpleSolver}}`` (src/ACME.jl:150).  On the GPU the ``CachingSolver``
    (src/solvers.jl:303-405: a per-stream, unboundedly growing k-d tree of stored solutions that only
    changes Newton's start point) keeps the last 16 stored solutions per instance (same lookup and
    storing rules; converged results agree within the solver tolerance, iteration counts follow the
    oracle's bounded variant).  ``solver=HomotopySolver`` / ``SimpleSolver`` select the cache-less stacks.
    """

    def __init__(self, circ=None, t=None, solver=CachingHomotopySolver, decompose_nonlinearity=True,
                 _data=None):
        if solver not in SOLVER_IDS:
            raise ValueError(f"unknown solver {solver!r}")
        self.solver = solver
        if _data is None:
            if isinstance(t, float):
                t = Fraction(t)   # exact binary value, like Rational{BigInt}(t)
            _data = _derive.derive(circ, Fraction(t), decompose_nonlinearity)
        d = _data
        self.nx, self.nu, self.ny = d["nx"], d["nu"], d["ny"]
        nnt = sum(d["nns"])
        self.nn_total = nnt
        nx, nu, ny = self.nx, self.nu, self.ny
        self.a, self.b, self.c = _f(d["a"], nx, nx), _f(d["b"], nx, nu), _f(d["c"], nx, nnt)
        self.x0 = np.asarray(d["x0"], dtype=np.float64).reshape(nx)
        self.dy, self.ey, self.fy = _f(d["dy"], ny, nx), _f(d["ey"], ny, nu), _f(d["fy"], ny, nnt)
        self.y0 = np.asarray(d["y0"], dtype=np.float64).reshape(ny)
        self.subs = []
        for k in range(d["nsub"]):
            nn, nq, np_ = d["nns"][k], d["nqs"][k], d["nps"][k]
            self.subs.append(SubProblem(
                nn, nq, np_, _f(d["pexps"][k], nq, np_), _f(d["dqs"][k], np_, nx),
                _f(d["eqs"][k], np_, nu), _f(d["fqprevs"][k], np_, nnt), _f(d["fqs"][k], nq, nn),
                d["q0s"][k], d["init_zs"][k], d["tables"][k],
                (d.get("row_orders") or [None] * d["nsub"])[k]))
        # mutable state (src/ACME.jl:137,145): starts at zero
        self.x = np.zeros(nx)

    # size accessors (src/ACME.jl:466-472), sub-problem index 1-based like the reference
    def np(self, k): return self.subs[k - 1].np
    def nq(self, k): return self.subs[k - 1].nq
    def nn(self, k=None):
        return self.nn_total if k is None else self.subs[k - 1].nn

    # --- (de)serialisation: small JSON fixtures --------------------------------------
    def to_dict(self):
        def m(a): return np.asarray(a).tolist()
        return dict(
            solver=self.solver, nx=self.nx, nu=self.nu, ny=self.ny, nsub=len(self.subs),
            nns=[s.nn for s in self.subs], nqs=[s.nq for s in self.subs],
            nps=[s.np for s in self.subs],
            a=m(self.a), b=m(self.b), c=m(self.c), x0=m(self.x0),
            dy=m(self.dy), ey=m(self.ey), fy=m(self.fy), y0=m(self.y0),
            pexps=[m(s.pexp) for s in self.subs], dqs=[m(s.dq) for s in self.subs],
            eqs=[m(s.eq) for s in self.subs], fqprevs=[m(s.fqprev) for s in self.subs],
            fqs=[m(s.fq) for s in self.subs], q0s=[m(s.q0) for s in self.subs],
            init_zs=[m(s.init_z) for s in self.subs], tables=[s.table for s in self.subs],
            row_orders=[s.row_order for s in self.subs])

    def tune_row_order(self, u=None, T=256, fs=44100):
        """Performance hint for the GPU kernel (no effect on results): find each sub-problem's
        usual LU pivot order with a short host-side pilot solve and store it as ``row_order``.
        ``u``: optional (nu, T) pilot input; default 1 kHz unit sine on input 1, every other
        input held at 0.5."""
        from . import hostsolve as hs
        import collections
        if not self.subs:
            return self
        if u is None:
            u = np.full((self.nu, T), 0.5)
            if self.nu:
                u[0] = np.sin(2 * np.pi * 1000.0 / fs * np.arange(T))
        if len(self.subs) != 1:
            return self
        s = self.subs[0]
        nleq = hs.HostNleq(s.table, s.fq.tolist(), s.q0.tolist(), s.pexp.tolist())
        seqs = collections.Counter()
        plain = hs.lu_factor

        def spy(A):
            r = plain(A)
            if r is not None:
                perm = list(range(len(A)))
                for k, kp in enumerate(r[1]):
                    perm[k], perm[kp] = perm[kp], perm[k]
                seqs[tuple(perm)] += 1
            return r
        hs.lu_factor = spy
        try:
            solver = hs.HostHomotopySolver(nleq, [0.0] * s.np, s.init_z.tolist())
            x = np.zeros(self.nx)
            for n in range(u.shape[1]):
                p = s.dq @ x + s.eq @ u[:, n]
                z = np.array(solver.solve(p.tolist()))
                if not np.isfinite(z).all():
                    break
                x = self.x0 + self.a @ x + self.b @ u[:, n] + self.c @ z
        finally:
            hs.lu_factor = plain
        if seqs:
            s.row_order = list(seqs.most_common(1)[0][0])
        return self

    @classmethod
    def from_dict(cls, d, solver=None):
        return cls(solver=solver or d.get("solver", CachingHomotopySolver), _data=d)

    def save(self, path):
        with open(path, "w") as fh:
            json.dump(self.to_dict(), fh)

    @classmethod
    def load(cls, path, solver=None):
        with open(path) as fh:
            return cls.from_dict(json.load(fh), solver)
