"""steadystate / steadystate! / linearize on the batched GPU solver (src/ACME.jl:474-550).

The reference solves, per sub-problem, a *derived* nonlinear equation whose q-offset and fq
fold the steady-state condition x = a x + b u + c z + x0 into the junction equations, with a
fresh ``HomotopySolver{SimpleSolver}`` started at (p = 0, z = 0) and tolerance 1e-15, then
recovers x from (I - a) x = b u + c z + x0.  Here the derived equation is packed as a model with
a single scalar parameter (pexp = steady_q0 as one column, q0 = 0): solving it at p = 1 from the
origin p = 0 walks exactly the reference's homotopy path pa = a * steady_q0.  The nonlinear
solve runs on the GPU through ``ModelRunner.solve`` (acme_batch_solve); the small dense linear
algebra around it is host-side numpy, as it is LAPACK in the reference (``lu(I - a)``).  Models
with several nonlinear sub-problems are handled like in the reference: one sub-problem after the
other.  ``initial_solution`` (src/ACME.jl:453-464) is the same kind of solve: ``solve_rays``.
"""
from __future__ import annotations

import numpy as np

from .model import DiscreteModel, HomotopySolver
from .runner import AcmeError, ModelRunner


def solve_only_model(table, row_order, nn, nq, q_target, fq, solver=HomotopySolver):
    """The reference's three-argument ``ParametricNonLinEq(func, nn, nq)`` (p == q, src/solvers.jl:
    23-28) restricted to the ray the homotopy walks: a model without states whose single scalar
    parameter s scales ``q_target`` (pexp = q_target as one column, q0 = 0, origin (s, z) = (0, 0)).
    ``solve`` at s = 1 then visits exactly the points pa = a * q_target of
    ``solve(HomotopySolver{SimpleSolver}(nleq, zeros(nq), zeros(nn)), q_target)`` --
    ``initial_solution`` (src/ACME.jl:453-464) and ``steadystate`` (:474-503) are both this."""
    d = dict(nx=0, nu=0, ny=0, nsub=1, nns=[nn], nqs=[nq], nps=[1],
             a=[], b=[], c=[], x0=[], dy=[], ey=[], fy=[], y0=[],
             pexps=[np.asarray(q_target, dtype=np.float64).reshape(nq, 1)], dqs=[np.zeros((1, 0))],
             eqs=[np.zeros((1, 0))], fqprevs=[np.zeros((1, nn))], fqs=[np.asarray(fq, dtype=np.float64)],
             q0s=[np.zeros(nq)], init_zs=[np.zeros(nn)], tables=[table],
             row_orders=[row_order])
    return DiscreteModel(solver=solver, _data=d)


def solve_rays(table, row_order, nn, nq, q_targets, fqs, tol=None, lib=None, device=None):
    """N equations res(q_target_i + fq_i z) = 0 solved on the GPU by homotopy from q = 0
    (``acme_batch_solve`` on per-instance ``solve_only_model``s; a single shared model when all
    instances have the same fq and q_target).  q_targets: (N, nq); fqs: (nq, nn) shared or
    (N, nq, nn).  Returns (z (N, nn), converged (N,))."""
    q_targets = np.atleast_2d(np.asarray(q_targets, dtype=np.float64))
    N = q_targets.shape[0]
    fqs = np.asarray(fqs, dtype=np.float64)
    per = fqs.ndim == 3 or N > 1
    models = [solve_only_model(table, row_order, nn, nq, q_targets[i], fqs[i] if fqs.ndim == 3 else fqs)
              for i in range(N)]
    r = ModelRunner(models[0], N, models=models if per and N > 1 else None, lib=lib, device=device)
    if tol is not None:
        r.set_resabstol(tol)
    z, conv, _ = r.solve(np.ones((N, 1)))
    return z, conv


def steadystate(model, u=None, lib=None, device=None):
    """``steadystate(model, u)`` (src/ACME.jl:474-503): the state x with x = a x + b u + c z(x, u) + x0.

    ``u``: (nu,) -> returns (nx,); or (N, nu) for N different operating points -> (N, nx) (batched
    GPU solves with per-instance equations).  The sub-problems are solved one after another like
    in the reference, each seeing the steady z of the earlier ones through fqprev and through the
    state (dq (I - a)^-1 c).  Raises like the reference if a solver does not converge."""
    single = u is None or np.ndim(u) == 1
    U = np.zeros((1, model.nu)) if u is None else np.atleast_2d(np.asarray(u, dtype=np.float64))
    N = U.shape[0]
    IA = np.eye(model.nx) - model.a
    solve = (lambda rhs: np.linalg.solve(IA, rhs)) if model.nx else (lambda rhs: rhs)
    IAinv = np.linalg.inv(IA) if model.nx else np.zeros((0, 0))
    steady_z = np.zeros((N, model.nn_total))
    zoff = 0
    for s in model.subs:
        dqIA = s.dq @ IAinv if model.nx else np.zeros((s.np, 0))           # dq / IA_LU
        zcols = slice(zoff, zoff + s.nn)
        fq_s = s.pexp @ dqIA @ model.c[:, zcols] + s.fq
        gain_u = dqIA @ model.b + s.eq
        gain_z = dqIA @ model.c + s.fqprev
        q_targets = np.stack([s.q0 + s.pexp @ (gain_u @ U[i] + gain_z @ steady_z[i]) + s.pexp @ (dqIA @ model.x0)
                              for i in range(N)])
        z, conv = solve_rays(s.table, s.row_order, s.nn, s.nq, q_targets, fq_s, tol=1e-15, lib=lib, device=device)
        if not conv.all():
            raise AcmeError("Failed to find steady state solution")
        steady_z[:, zcols] = z
        zoff += s.nn
    X = np.stack([solve(model.b @ U[i] + model.c @ steady_z[i] + model.x0) for i in range(N)])
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


def linearize(model, usteady=None, lib=None, device=None, reference_offsets=None):
    """``linearize(model, usteady)`` (src/ACME.jl:505-550, src/solvers.jl:407-414): the small-signal
    linear ``DiscreteModel`` around the steady state for the constant input ``usteady``.

    The two nonlinear solves (steady state at tolerance 1e-15, then ``solve(solver, psteady)`` with
    the model's own solver settings) and ``get_extrapolation_jacobian`` = -J \\ Jp at the solution
    (``acme_batch_get_extrapolation_jacobian``) run on the GPU; the assembly of the linear model is a
    handful of small dense products on the host, as they are BLAS calls in the reference.

    Models with several nonlinear sub-problems: the small-signal gains (a, b, dy, ey) follow the
    reference's recursion (dqlins / eqlins, :530-532).  For the constant terms (x0, y0) the
    reference adds c (z_k - dzdp_k p_k) per sub-problem (:534,538) -- the DEFAULT here
    (``reference_offsets=True``: upstream results are reproduced literally).  That formula leaves out
    what the affine offsets of the EARLIER sub-problems contribute to p_k through fqprev: the linear
    model of a decomposed circuit then does not reproduce the operating point it was linearised at
    (0.05 V off on tests/circuits.two_stage_clipper; the reference's own tests only linearise
    single-sub-problem models, where the two forms coincide).  ``reference_offsets=False`` carries the
    offsets through, which makes decomposed and non-decomposed derivations of one circuit agree to
    rounding."""
    if reference_offsets is None:
        reference_offsets = True
        if len(model.subs) > 1:
            import warnings
            warnings.warn("linearize: model has several nonlinear sub-problems; the reference's constant terms "
                          "(src/ACME.jl:534,538) are reproduced literally and miss the operating point -- pass "
                          "reference_offsets=False to carry the earlier sub-problems' offsets through",
                          stacklevel=2)
    u = np.zeros(model.nu) if usteady is None else np.asarray(usteady, dtype=np.float64)
    xs = steadystate(model, u, lib=lib, device=device)
    x0, a, b = model.x0.copy(), model.a.copy(), model.b.copy()
    y0, dy, ey = model.y0.copy(), model.dy.copy(), model.ey.copy()
    if model.subs:
        r = ModelRunner(model, 1, lib=lib, device=device)
        zsteady = np.zeros(model.nn_total)
        zranges, dzdps, dqlins, eqlins, offs = [], [], [], [], []
        zoff = 0
        for idx, s in enumerate(model.subs):
            ps = s.dq @ xs + s.eq @ u + s.fqprev @ zsteady
            z, conv, _ = r.solve(ps[None, :], sub=idx)             # linearize(solver, psteady), src/solvers.jl:407-414
            if not conv.all():
                raise ValueError(f"Cannot linearize because no solution found at p={ps}")
            z = z[0]
            # set_extrapolation_origin(solver, p, z) (src/solvers.jl:409): explicit, as in the reference -- a
            # solve that reported convergence through the singular-J / small-residual exit has NOT moved it
            _, p_all, z_all = r.get_state()
            po = sum(t.np for t in model.subs[:idx])
            p_all[0, po:po + s.np] = ps
            z_all[0, zoff:zoff + s.nn] = z
            r.set_state(p=p_all, z=z_all)
            dzdp = r.get_extrapolation_jacobian(sub=idx)[0]        # -(J \ Jp) at (ps, z)
            if not np.isfinite(dzdp).all():
                raise ValueError(f"Cannot linearize: singular Jacobian at p={ps}")
            zr = slice(zoff, zoff + s.nn)
            zsteady[zr] = z
            fqdzdps = [s.fqprev[:, zranges[n]] @ dzdps[n] for n in range(idx)]
            dqlin = s.dq + sum((f @ d for f, d in zip(fqdzdps, dqlins)), np.zeros_like(s.dq))
            eqlin = s.eq + sum((f @ e for f, e in zip(fqdzdps, eqlins)), np.zeros_like(s.eq))
            # z_k ~ off_k + dzdp_k (dqlin_k x + eqlin_k u): the constant part of p_k is what the earlier
            # sub-problems' offsets feed in through fqprev
            pconst = sum((s.fqprev[:, zranges[n]] @ offs[n] for n in range(idx)), np.zeros(s.np))
            off = z - dzdp @ (ps - (0.0 if reference_offsets else pconst))
            zranges.append(zr); dzdps.append(dzdp); dqlins.append(dqlin); eqlins.append(eqlin); offs.append(off)
            x0 += model.c[:, zr] @ off
            a += model.c[:, zr] @ dzdp @ dqlin
            b += model.c[:, zr] @ dzdp @ eqlin
            y0 += model.fy[:, zr] @ off
            dy += model.fy[:, zr] @ dzdp @ dqlin
            ey += model.fy[:, zr] @ dzdp @ eqlin
            zoff += s.nn
    nx, nu, ny = model.nx, model.nu, model.ny
    d = dict(nx=nx, nu=nu, ny=ny, nsub=0, nns=[], nqs=[], nps=[], a=a, b=b, c=np.zeros((nx, 0)), x0=x0,
             dy=dy, ey=ey, fy=np.zeros((ny, 0)), y0=y0, pexps=[], dqs=[], eqs=[], fqprevs=[], fqs=[], q0s=[],
             init_zs=[], tables=[])
    return DiscreteModel(solver=model.solver, _data=d)
