"""steadystate / steadystate! on the batched GPU solver (src/ACME.jl:474-503).

The reference solves, per sub-problem, a *derived* nonlinear equation whose q-offset and fq
fold the steady-state condition x = a x + b u + c z + x0 into the junction equations, with a
fresh ``HomotopySolver{SimpleSolver}`` started at (p = 0, z = 0) and tolerance 1e-15, then
recovers x from (I - a) x = b u + c z + x0.  Here the derived equation is packed as a model with
a single scalar parameter (pexp = steady_q0 as one column, q0 = 0): solving it at p = 1 from the
origin p = 0 walks exactly the reference's homotopy path pa = a * steady_q0.  The nonlinear
solve runs on the GPU through ``ModelRunner.solve`` (acme_batch_solve); the small dense linear
algebra around it is host-side numpy, as it is LAPACK in the reference (``lu(I - a)``).
"""
from __future__ import annotations

import numpy as np

from .model import DiscreteModel, HomotopySolver
from .runner import AcmeError, ModelRunner


def _derived_model(model, steady_q0, fq_s):
    s = model.subs[0]
    d = dict(nx=0, nu=0, ny=0, nsub=1, nns=[s.nn], nqs=[s.nq], nps=[1],
             a=[], b=[], c=[], x0=[], dy=[], ey=[], fy=[], y0=[],
             pexps=[np.asarray(steady_q0).reshape(s.nq, 1)], dqs=[np.zeros((1, 0))],
             eqs=[np.zeros((1, 0))], fqprevs=[np.zeros((1, s.nn))], fqs=[fq_s],
             q0s=[np.zeros(s.nq)], init_zs=[np.zeros(s.nn)], tables=[s.table],
             row_orders=[s.row_order])
    return DiscreteModel(solver=HomotopySolver, _data=d)


def steadystate(model, u=None, lib=None, device=None):
    """``steadystate(model, u)``: the state x with x = a x + b u + c z(x, u) + x0.

    ``u``: (nu,) -> returns (nx,); or (N, nu) for N different operating points -> (N, nx)
    (one batched GPU solve with per-instance equations).  Raises like the reference if the
    solver does not converge."""
    if len(model.subs) > 1:
        raise AcmeError("steadystate on the GPU supports a single nonlinear sub-problem")
    single = u is None or np.ndim(u) == 1
    U = np.zeros((1, model.nu)) if u is None else np.atleast_2d(np.asarray(u, dtype=np.float64))
    N = U.shape[0]
    IA = np.eye(model.nx) - model.a
    solve = (lambda rhs: np.linalg.solve(IA, rhs)) if model.nx else (lambda rhs: rhs)
    if not model.subs:
        X = np.stack([solve(model.b @ U[i] + model.x0) for i in range(N)])
        return X[0] if single else X
    s = model.subs[0]
    dqIA = s.dq @ np.linalg.inv(IA) if model.nx else np.zeros((s.np, 0))   # dq / IA_LU
    fq_s = s.pexp @ dqIA @ model.c + s.fq
    derived = []
    for i in range(N):
        steady_q0 = s.q0 + s.pexp @ ((dqIA @ model.b + s.eq) @ U[i]) + s.pexp @ (dqIA @ model.x0)
        derived.append(_derived_model(model, steady_q0, fq_s))
    r = ModelRunner(derived[0], N, models=derived if N > 1 else None, lib=lib, device=device)
    r.set_resabstol(1e-15)
    z, conv, _ = r.solve(np.ones((N, 1)))
    if not conv.all():
        raise AcmeError("Failed to find steady state solution")
    X = np.stack([solve(model.b @ U[i] + model.c @ z[i] + model.x0) for i in range(N)])
    return X[0] if single else X


def steadystate_(runner, u=None):
    """``steadystate!(model, u)``: put every instance of ``runner`` at its steady state
    (``u``: (nu,) shared or (N, nu)).  Returns the steady states (N, nx)."""
    m = runner.model
    U = np.zeros(m.nu) if u is None else np.asarray(u, dtype=np.float64)
    if U.ndim == 1:
        X = np.tile(steadystate(m, U, lib=runner.lib), (runner.n, 1))
    else:
        X = steadystate(m, U, lib=runner.lib)
    runner.set_state(x=X)
    return X
