"""The one parity case held to RTOL instead of RTOL_SAME, explained (VERDICT r5 "what's weak" 2): on the 18-instance `birdie_var`
sweep the kernels need 8 % more Newton iterations than the oracle -- in the four loudest instances only -- with outputs
equal to 1.9e-9.  Shared by the emulator test (tests/test_emu_parity.py) and its GPU twin (tests/test_gpu_parity.py).

What happens: at a few samples per period the input slews so fast that the extrapolated start of the direct attempt
(src/solvers.jl:209-215) lands where the transistor's junction exponentials are ~1e16: the Jacobian has entries of 1e17
beside entries of 1, a condition number beyond 1 / eps, and the first Newton step is decided by ROUNDING.  The reference's
arithmetic steps from z[0] = -9.93 to -34 932, then to 9.5e6, overflows (src/solvers.jl:219-221: solve returns, not
converged) and the homotopy takes over: 35 iterations for the sample.  The kernels' arithmetic -- fused multiply-adds, a
1-ulp exp, on the 16-lane kernels another elimination order -- steps to -9.931 from the same point and the direct attempt
converges by itself after 51 iterations.  Both end at the same z (5e-12).  Neither is "the" path: the ORACLE's own
iteration count at such a sample changes with a 1-ulp change of p, and whether its direct attempt converges flips."""
import numpy as np

NAME, N, T = "birdie_var", 18, 2048
PARTING = {14: 40, 17: 2}                     # instance -> first sample whose iteration counts differ (instances 0 ... 13: none)


def per_sample_iterations(make_runner, model, u_inst, upto):
    """iterations of every sample < upto of one instance on the oracle and on a kernel runner (one launch per sample)"""
    from oracle.refpy import RefRunner
    ref, r = RefRunner(model, None), make_runner(1)
    io, ik, done = [], [], 0
    for n in range(upto):
        ref.run(u_inst[:, n:n + 1])
        r.run(u_inst[None, :, n:n + 1])
        io.append(int(ref.report.iters_total))
        tot = int(r.report_arrays()["iters_total"][0])
        ik.append(tot - done)
        done = tot
    return np.array(io), np.array(ik)


def direct_attempts(make_runner, model, u_inst, n0):
    """the state both sides share before sample n0, and what each side's DIRECT attempt (SimpleSolver from the origin) does
    with sample n0's p: (oracle: converged, iterations), (kernel: converged, iterations), |z difference| of the full solves"""
    from helpers import load
    from oracle.refpy import RefRunner
    ref, r = RefRunner(model, None), make_runner(1)
    if n0:
        ref.run(u_inst[:, :n0])
        r.run(u_inst[None, :, :n0])
    xo = ref.x
    po, zo = ref.get_origin(0)
    xg, pg, zg = r.get_state()
    assert np.abs(xo - xg[0]).max() < 1e-12 and np.abs(zo - zg[0]).max() < 1e-9
    s = model.subs[0]
    p = np.array(s.dq, dtype=float) @ xo + np.array(s.eq, dtype=float) @ u_inst[:, n0]
    out = {}
    for solver in ("SimpleSolver", "HomotopySolver{SimpleSolver}"):
        ro = RefRunner(model, solver)
        ro.x = xo
        ro.set_origin(0, po, zo)
        z, conv, its = ro.solve(p)
        m2 = load(NAME)
        m2.solver = solver
        rg = make_runner(1, model=m2)
        rg.set_state(x=xg, p=pg, z=zg)
        zk, ck, ik = rg.solve(p[None, :])
        out[solver] = ((bool(conv), int(its)), (bool(ck[0]), int(ik[0])), float(np.abs(np.asarray(z) - np.asarray(zk)[0]).max()))
    return (xo, po, zo, p), out


def oracle_under_ulps(model, state, span=4):
    """the ORACLE's direct attempt and full solve with p[0] moved by -span ... +span ulps: [(ulps, direct converged, its, its of the full solve)]"""
    from oracle.refpy import RefRunner
    xo, po, zo, p = state
    rows = []
    for k in range(-span, span + 1):
        pk = p.copy()
        for _ in range(abs(k)):
            pk[0] = np.nextafter(pk[0], np.inf if k > 0 else -np.inf)
        res = []
        for solver in ("SimpleSolver", "HomotopySolver{SimpleSolver}"):
            ro = RefRunner(model, solver)
            ro.x = xo
            ro.set_origin(0, po, zo)
            _, conv, its = ro.solve(pk)
            res += [bool(conv), int(its)]
        rows.append((k, res[0], res[1], res[3]))
    return rows
