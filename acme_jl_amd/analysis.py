"""steadystate / steadystate! / linearize on the batched GPU solver (src/ACME.jl:474-550).

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


def linearize(model, usteady=None, lib=None, device=None):
    """``linearize(model, usteady)`` (src/ACME.jl:505-550, src/solvers.jl:407-414): the small-signal
    linear ``DiscreteModel`` around the steady state for the constant input ``usteady``.

    The two nonlinear solves (steady state at tolerance 1e-15, then ``solve(solver, psteady)`` with
    the model's own solver settings) and ``get_extrapolation_jacobian`` = -J \\ Jp at the solution
    (``acme_batch_get_extrapolation_jacobian``) run on the GPU; the assembly of the linear model is a
    handful of small dense products on the host, as they are BLAS calls in the reference."""
    if len(model.subs) > 1:
        raise AcmeError("linearize on the GPU supports a single nonlinear sub-problem")
    u = np.zeros(model.nu) if usteady is None else np.asarray(usteady, dtype=np.float64)
    xs = steadystate(model, u, lib=lib, device=device)
    x0, a, b = model.x0.copy(), model.a.copy(), model.b.copy()
    y0, dy, ey = model.y0.copy(), model.dy.copy(), model.ey.copy()
    if model.subs:
        s = model.subs[0]
        ps = s.dq @ xs + s.eq @ u
        r = ModelRunner(model, 1, lib=lib, device=device)
        z, conv, _ = r.solve(ps[None, :])
        if not conv.all():
            raise ValueError(f"Cannot linearize because no solution found at p={ps}")
        z = z[0]
        dzdp = r.get_extrapolation_jacobian()[0]       # -(J \ Jp) at (ps, z): the solve just made it the origin
        if not np.isfinite(dzdp).all():
            raise ValueError(f"Cannot linearize: singular Jacobian at p={ps}")
        x0 += model.c @ (z - dzdp @ ps)
        a += model.c @ dzdp @ s.dq
        b += model.c @ dzdp @ s.eq
        y0 += model.fy @ (z - dzdp @ ps)
        dy += model.fy @ dzdp @ s.dq
        ey += model.fy @ dzdp @ s.eq
    nx, nu, ny = model.nx, model.nu, model.ny
    d = dict(nx=nx, nu=nu, ny=ny, nsub=0, nns=[], nqs=[], nps=[], a=a, b=b, c=np.zeros((nx, 0)), x0=x0,
             dy=dy, ey=ey, fy=np.zeros((ny, 0)), y0=y0, pexps=[], dqs=[], eqs=[], fqprevs=[], fqs=[], q0s=[],
             init_zs=[], tables=[])
    return DiscreteModel(solver=model.solver, _data=d)
