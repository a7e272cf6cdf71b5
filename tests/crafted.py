"""Hand-crafted DiscreteModels that expose single pieces of the solver stack through the C ABI
(acme_batch_solve), so that the HIP path's LinearSolver / SimpleSolver / HomotopySolver can be
pinned at unit level against the oracle -- the reference does this with closures
(test/runtests.jl:23-41, 207-219), which cannot cross a C ABI; here the same equations are
assembled from element-table rows.

  linear_system_model(A)   res(z) = A z - p      (potentiometer rows with pos, i pinned to 0)
  parabola_model()         res(z) = z^2 - 1 + p  (one MOSFET in saturation: alpha/2 = 1, vt = 0)
"""
import numpy as np


def _model(nn, nq, np_, pexp, fq, q0, init_z, table, solver=None):
    from acme_jl_amd.model import DiscreteModel, HomotopySolver
    d = dict(nx=0, nu=0, ny=0, nsub=1, nns=[nn], nqs=[nq], nps=[np_],
             a=[], b=[], c=[], x0=[], dy=[], ey=[], fy=[], y0=[],
             pexps=[np.asarray(pexp, dtype=float).reshape(nq, np_)], dqs=[np.zeros((np_, 0))],
             eqs=[np.zeros((np_, 0))], fqprevs=[np.zeros((np_, nn))],
             fqs=[np.asarray(fq, dtype=float).reshape(nq, nn)], q0s=[np.asarray(q0, dtype=float)],
             init_zs=[np.asarray(init_z, dtype=float)], tables=[table], row_orders=[None])
    return DiscreteModel(solver=solver or HomotopySolver, _data=d)


def linear_system_model(A, solver=None, B=None, w=None):
    """nn = A.shape[0] (even): the nonlinear equation is the LINEAR system  A z - p = 0, built from
    nn/2 potentiometers (src/elements.jl:25-30: res = [v1 - r pos i1, v2 - r (1-pos) i2],
    q = (v1, v2, i1, i2, pos), r = 1) whose currents and position are pinned to 0 (zero rows of
    fq, pexp and q0), so that row 2k of the element table is v1 = (A z - p)[2k] and row 2k+1 is
    v2.  The first-order extrapolated start (src/solvers.jl:209-215) is then already exact: z
    comes out of the origin's J^-1 Jp, needediterations = 1.

    With ``B`` (nn/2 x nn) and ``w`` (nn,) the wiper position becomes pos = w.p and the current
    i1 = B[k].z:  row 2k reads  (A[2k] - (w.p) B[k]) z - p[2k] = 0 -- still linear in z for a given
    p, but bilinear in (p, z), so the extrapolation from (0, 0) misses and exactly ONE Newton step
    with the Jacobian M(p) = A - (w.p) [B[0]; 0; B[1]; 0; ...] follows (needediterations = 2):
    that step is the plain elimination J dz = res of the hot loop."""
    A = np.asarray(A, dtype=float)
    nn = A.shape[0]
    assert A.shape == (nn, nn) and nn % 2 == 0
    npots = nn // 2
    nq = 5 * npots
    fq = np.zeros((nq, nn))
    pexp = np.zeros((nq, nn))
    table = []
    for k in range(npots):
        fq[5 * k + 0] = A[2 * k]
        fq[5 * k + 1] = A[2 * k + 1]
        pexp[5 * k + 0, 2 * k] = -1.0
        pexp[5 * k + 1, 2 * k + 1] = -1.0
        if B is not None:
            fq[5 * k + 2] = np.asarray(B, dtype=float)[k]
            pexp[5 * k + 4] = np.asarray(w, dtype=float)
        table.append(dict(kind=3, par=[1.0], nq=5, nn=2, qoff=5 * k, roff=2 * k))
    return _model(nn, nq, nn, pexp, fq, np.zeros(nq), np.zeros(nn), table, solver)


def bilinear_matrix(A, B, w, p):
    """M(p) of linear_system_model(A, B=B, w=w): the Jacobian of its one Newton step."""
    M = np.array(A, dtype=float)
    M[0::2] -= float(np.dot(w, p)) * np.asarray(B, dtype=float)
    return M


def parabola_model(solver=None):
    """test/runtests.jl:207-219: res = z^2 - 1 + p, J = 2z, extrapolation origin (p, z) = (0, 1).
    One MOSFET (src/elements.jl:453-479) in saturation, q = (vgs, vds, id) = (z, 10, 1 - p),
    alpha = (2,), vt = (0,), lambda = 0:  res = alpha/2 (vgs - vt)^2 - id = z^2 - 1 + p.  For
    z <= 0 the element is cut off (res = p - 1, J = 0): like the parabola, no root for p > 1."""
    par = [1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 2.0, 0.0, 0.0, 0.0]
    table = [dict(kind=4, par=par, nq=3, nn=1, qoff=0, roff=0)]
    return _model(1, 3, 1, [[0.0], [0.0], [-1.0]], [[1.0], [0.0], [0.0]], [0.0, 10.0, 1.0], [1.0], table, solver)
