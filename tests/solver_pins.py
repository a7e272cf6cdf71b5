"""Unit-level pins of LinearSolver / SimpleSolver / HomotopySolver on a library implementing
include/acme_hip.h -- bodies shared by the -m gpu tests (libacme_hip.so) and the GPU-less
emulator tests (the same kernel source on the CPU wave emulator).  The reference pins these with
closures (test/runtests.jl:23-41, 207-219); here the same equations come from crafted element
tables (tests/crafted.py) and are driven through acme_batch_solve."""
import numpy as np

from crafted import bilinear_matrix, linear_system_model, parabola_model


def _runner(lib, model, n, **kw):
    from acme_jl_amd.runner import ModelRunner
    return ModelRunner(model, n, lib=lib, **kw)


def embed(A3):
    """3x3 -> 4x4 (potentiometer rows come in pairs): one decoupled unit row."""
    A = np.eye(4)
    A[:3, :3] = A3
    return A


def check_reference_lu_cases(lib):
    """test/runtests.jl:23-41: A = [1 .5 .4; 2 4 1.7; 4 7 9.1] solves (A*y ~ x, in place too) and
    zeros(3,3) is reported singular.  On the HIP path the LinearSolver is the in-register
    elimination inside solve(): the linear system A z = p must come out of the extrapolation from
    the origin (J^-1 Jp, needediterations = 1) -- and, with the bilinear variant of the crafted
    model, out of exactly one plain Newton step (needediterations = 2) -- to 1e-13 of the oracle's
    setlhs!/solve! restatement; an all-zero Jacobian must end as not converged."""
    from oracle.refpy import lu_factor, lu_solve
    A3 = np.array([[1.0, 0.5, 0.4], [2.0, 4.0, 1.7], [4.0, 7.0, 9.1]])
    rng = np.random.default_rng(11)
    X = rng.random((5, 3))
    r = _runner(lib, linear_system_model(embed(A3)), 5)
    z, conv, its = r.solve(np.concatenate([X, np.zeros((5, 1))], axis=1))
    ok, f, ipiv = lu_factor(A3)
    assert ok and conv.all() and (its == 1).all(), (conv, its)
    for i in range(5):
        yref = lu_solve(f, ipiv, X[i])
        np.testing.assert_allclose(z[i, :3], yref, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(A3 @ z[i, :3], X[i], rtol=1e-13)
        assert z[i, 3] == 0.0
    # the plain Newton-step elimination: bilinear variant, one step with M(p) = A - (w.p) B
    B = rng.standard_normal((2, 4)) * 0.3
    w = np.array([0.5, -0.25, 0.125, 0.0])
    P = np.concatenate([X, np.zeros((5, 1))], axis=1)
    rb = _runner(lib, linear_system_model(embed(A3), B=B, w=w), 5)
    zb, convb, itsb = rb.solve(P)
    assert convb.all() and (itsb == 2).all(), (convb, itsb)
    for i in range(5):
        ok, f, ipiv = lu_factor(bilinear_matrix(embed(A3), B, w, P[i]))
        np.testing.assert_allclose(zb[i], lu_solve(f, ipiv, P[i]), rtol=1e-13, atol=1e-15)
    # singular: !setlhs!(solver, zeros(3,3))
    assert not lu_factor(np.zeros((3, 3)))[0]
    rs = _runner(lib, linear_system_model(np.zeros((4, 4))), 2)
    _, conv, _ = rs.solve(np.array([[1.0, 2.0, 3.0, 4.0], [0.5, 0.0, 0.0, 0.0]]))
    assert not conv.any()
    # rank-deficient but not zero (pivot search finds an exact zero later)
    B = np.array([[1.0, 2.0, 3.0, 0], [2.0, 4.0, 6.0, 0], [1.0, 0.0, 1.0, 0], [0, 0, 0, 1.0]])
    assert not lu_factor(B)[0]
    _, conv, _ = _runner(lib, linear_system_model(B), 1).solve(np.array([[1.0, 1.0, 1.0, 1.0]]))
    assert not conv.any()


def pivot_sweep_matrices(n, count, seed):
    """Matrices on which threshold pivoting (|l| <= 8 keeps the row in place) and the reference's
    strict first-maximum rule choose DIFFERENT row orders, plus ones that force interchanges on
    either rule, plus plain well-conditioned ones."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        A = rng.standard_normal((n, n))
        kind = k % 4
        if kind == 0:          # diagonal modest, sub-diagonal entries 1..3.5x larger: strict swaps, threshold keeps
            A = np.eye(n) + 0.1 * A
            for j in range(n - 1):
                A[rng.integers(j + 1, n), j] = rng.uniform(1.2, 3.5) * rng.choice([-1, 1])
        elif kind == 1:        # tiny diagonal: both rules must interchange
            A[np.arange(n), np.arange(n)] = 1e-9 * rng.standard_normal(n)
        elif kind == 2:        # a permuted diagonally dominant matrix
            A = (np.eye(n) * 5 + 0.3 * A)[rng.permutation(n)]
        out.append(A)
    return out


def check_pivot_sweep(lib, n=8, count=48, seed=5, per_instance=False):
    """A z = p for a batch of per-instance matrices: the HIP elimination (adopted row order,
    threshold pivoting, Gauss-Jordan) against the oracle's partially pivoted LU (first strict
    maximum, src/solvers.jl:58-132): |z - z_ref|_inf / |z_ref|_inf <= 1e-13 + 4 eps cond(M)."""
    from oracle.refpy import lu_factor, lu_solve
    mats = pivot_sweep_matrices(n, count, seed)
    rng = np.random.default_rng(seed + 1)
    P = rng.standard_normal((count, n))
    Bs = 0.2 * rng.standard_normal((count, n // 2, n))
    w = rng.standard_normal(n) / n
    worst = 0.0
    for bilinear in (False, True):
        models = [linear_system_model(A, **(dict(B=Bs[i], w=w) if bilinear else {})) for i, A in enumerate(mats)]
        if per_instance:     # one batch, every instance its own matrices (small shapes only: LDS)
            z, conv, its = _runner(lib, models[0], count, models=models).solve(P)
        else:                # one single-instance batch per matrix
            res = [_runner(lib, models[i], 1).solve(P[i:i + 1]) for i in range(count)]
            z, conv, its = (np.concatenate([r[k] for r in res]) for k in range(3))
        assert conv.all(), np.flatnonzero(~conv)
        for i, A in enumerate(mats):
            M = bilinear_matrix(A, Bs[i], w, P[i]) if bilinear else A
            ok, f, ipiv = lu_factor(M)
            assert ok
            zref = lu_solve(f, ipiv, P[i])
            err = np.abs(z[i] - zref).max() / max(np.abs(zref).max(), 1e-300)
            # two backward-stable eliminations of the same matrix: 1e-13, plus 4 eps cond(M) for the
            # few deliberately ill-conditioned cases (cond up to ~4e3)
            worst = max(worst, err / (1.0 + 4 * 2.2e-16 * np.linalg.cond(M) / 1e-13))
            # extrapolation alone / one step (+ one polishing step when the case is badly conditioned)
            assert its[i] <= (3 if bilinear else 2), (bilinear, i, its[i])
    assert worst < 1e-13, worst
    return worst


def check_parabola(lib):
    """test/runtests.jl:207-219 through acme_batch_solve: z^2 - 1 + p converges for p in
    [-0.5, 0.5) and must NOT converge for p >= 1.5 (no real root; the homotopy bisects down to
    adjacent floats and gives up, src/solvers.jl:286-290); z, hasconverged and needediterations
    against the oracle's HomotopySolver on the same equation."""
    from oracle.refpy import RefRunner
    m = parabola_model()
    rng = np.random.default_rng(2)
    p_ok = -0.5 + rng.random(12)
    p_bad = 1.5 + rng.random(4)
    p = np.concatenate([p_ok, p_bad])[:, None]
    r = _runner(lib, m, len(p))
    z, conv, its = r.solve(p)
    assert conv[:12].all() and not conv[12:].any(), conv
    np.testing.assert_allclose(z[:12, 0], np.sqrt(1 - p_ok), rtol=1e-10)
    for i in range(len(p)):
        zr, cr, ir = RefRunner(m).solve(p[i])
        assert cr == conv[i] and ir == its[i], (i, cr, conv[i], ir, its[i])
        if cr:
            np.testing.assert_allclose(z[i], zr, rtol=1e-13)
    # SimpleSolver alone: same roots where Newton from (0, 1) reaches them, failure where not
    from acme_jl_amd.model import SimpleSolver
    ms = parabola_model(SimpleSolver)
    zs, convs, _ = _runner(lib, ms, len(p)).solve(p)
    for i in range(len(p)):
        zr, cr, _ = RefRunner(ms).solve(p[i])
        assert cr == convs[i]


def check_extrapolation_jacobian(lib):
    """acme_batch_get_extrapolation_jacobian = get_extrapolation_jacobian(solver) (src/solvers.jl:
    198-201) = -(J \\ Jp) at each instance's extrapolation origin: against a host evaluation of the
    element table at the same (p, z), for the big (recorded-elimination) and the small (J^-1 Jp)
    origin representations, after solves that moved the origins to different points."""
    from helpers import load
    from acme_jl_amd.hostsolve import eval_table
    from test_gpu_parity import trajectory_ps
    from helpers import sweep_inputs
    for name in ("superover_var", "superover_fixed", "birdie_var", "diodeclipper"):
        m = load(name)
        s = m.subs[0]
        ps = trajectory_ps(m, sweep_inputs(name, 2, 300)[1], every=23)
        N = len(ps)
        r = _runner(lib, m, N)
        # fresh batch: origin = (0, init_z)
        jac0 = r.get_extrapolation_jacobian()
        z, conv, _ = r.solve(ps)
        assert conv.all()
        jac = r.get_extrapolation_jacobian()
        _, lp, lz = r.get_state()
        np.testing.assert_allclose(lp, ps, rtol=0, atol=0)
        for i in range(N):
            for (p_, z_, got) in ((np.zeros(s.np), s.init_z, jac0[i]), (lp[i], lz[i], jac[i])):
                q = s.q0 + s.pexp @ p_ + s.fq @ z_
                _, jq = eval_table(s.table, q.tolist(), s.nn, s.nq)
                jq = np.asarray(jq)
                ref = -np.linalg.solve(jq @ s.fq, jq @ s.pexp)
                np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
        # the export does not disturb the batch: the same solve again gives the same answer bit for bit
        z2, _, _ = _runner(lib, m, N).solve(ps)
        assert np.array_equal(z, z2)


def check_decomposed_analysis(lib):
    """steadystate / linearize on a model with several nonlinear sub-problems (src/ACME.jl:474-550):
    the decomposed two-stage clipper (nsub = 2, the second stage fed by the first through fqprev)
    against the non-decomposed derivation of the same circuit -- steady state equal to rounding and a
    fixed point of the oracle's run!, linear models equal to rounding and tracking the nonlinear model
    around the operating point; checksteady! of test/runtests.jl:664-671 on the decomposed model."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd.analysis import linearize, steadystate, steadystate_
    from acme_jl_amd.model import DiscreteModel
    from oracle.refpy import RefRunner
    t = Fraction(1, 44100)
    m = DiscreteModel(circuits.two_stage_clipper(0.6), t)
    m1 = DiscreteModel(circuits.two_stage_clipper(0.6), t, decompose_nonlinearity=False)
    assert len(m.subs) == 2 and len(m1.subs) == 1 and np.abs(m.subs[1].fqprev).max() > 0
    xs, xs1 = steadystate(m, lib=lib), steadystate(m1, lib=lib)
    np.testing.assert_allclose(xs, xs1, rtol=1e-9, atol=1e-18)
    # N operating points at once
    U = np.linspace(-0.3, 0.4, 5)[:, None]
    X = steadystate(m, U, lib=lib)
    for i in range(5):
        np.testing.assert_allclose(X[i], steadystate(m1, U[i], lib=lib), rtol=1e-8, atol=1e-16)
    # checksteady!: one sample from the steady state leaves the state where it is
    r = _runner(lib, m, 2)
    steadystate_(r)
    r.set_resabstol(1e-13)
    r.run(np.zeros((2, m.nu, 1)))
    np.testing.assert_allclose(r.get_state()[0], np.tile(xs, (2, 1)), rtol=1.5e-8, atol=1e-14)
    # (reference_offsets=False: the constant terms with the earlier sub-problems' offsets carried through)
    lin, lin1 = linearize(m, lib=lib, reference_offsets=False), linearize(m1, lib=lib)
    for k in ("a", "b", "x0", "dy", "ey", "y0"):
        np.testing.assert_allclose(getattr(lin, k), getattr(lin1, k), rtol=1e-6, atol=1e-9)
    u = 1e-5 * np.sin(2 * np.pi * 1000 / 44100 * np.arange(300))[None]
    ref = RefRunner(m)
    ref.set_resabstol(1e-14)
    ref.x = xs
    y = ref.run(u)
    rl = _runner(lib, lin, 1)
    steadystate_(rl)
    assert np.abs(rl.run(u) - y).max() < 5e-9          # second-order small at 10 uV
    # the reference's literal constant-term formula (the default) does not reproduce the operating point
    bad = linearize(m, lib=lib)
    assert np.abs(bad.y0 - lin1.y0).max() > 1e-3
