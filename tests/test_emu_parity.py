"""GPU-less logic tests: the *same kernel source* executed by the CPU wave emulator
(tests/emu) against the oracle.  These do not replace the -m gpu parity tests; they catch
logic errors (and cross-lane operations in divergent control flow) without a GPU."""
import numpy as np
import pytest

from helpers import assert_close, load, oracle_run, sweep_inputs


def emu_runner(emu_lib, model, n, **kw):
    from acme_jl_amd.runner import ModelRunner
    return ModelRunner(model, n, lib=emu_lib, **kw)


@pytest.mark.parametrize("name,N,T", [
    ("diodeclipper", 19, 200),
    ("superover_fixed", 5, 150),
    ("superover_var", 4, 100),
    ("birdie_fixed", 4, 200),
    ("birdie_var", 6, 200),
    ("rc_ladder", 3, 64),
])
def test_emulated_kernel_matches_oracle(emu_lib, name, N, T):
    m = load(name)
    u = sweep_inputs(name, N, T)
    r = emu_runner(emu_lib, m, N)
    y = r.run(u)
    yref, _ = oracle_run(m, u)
    assert_close(y, yref)


def test_emulated_state_roundtrip(emu_lib):
    m = load("birdie_fixed")
    u = sweep_inputs("birdie_fixed", 3, 120)
    y_once = emu_runner(emu_lib, m, 3).run(u)
    r = emu_runner(emu_lib, m, 3)
    y_split = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
    assert np.array_equal(y_once, y_split)
    x, p, z = r.get_state()
    assert x.shape == (3, m.nx) and p.shape == (3, 2) and z.shape == (3, 2)


def test_emulated_analytic_circuits(emu_lib):
    """Every element kind (diode, BJT Ebers-Moll + all Gummel-Poon branches, MOSFET, tanh
    op-amp, Jiles-Atherton core) through the padded generic shapes."""
    from helpers import analytic_cases
    for name, m, u in analytic_cases():
        r = emu_runner(emu_lib, m, u.shape[0])
        y = r.run(u)
        yref, its = oracle_run(m, u)
        rel = assert_close(y, yref)
        print(name, r.kernel_shape(), f"rel err {rel:.2e}")


def test_emulated_failure_semantics(emu_lib):
    """test/runtests.jl:170-183: unsolvable input -> warning + finite output; Inf input ->
    the reference throws; that instance stops, the others are unaffected."""
    import warnings
    from fractions import Fraction
    import circuits
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import AcmeError
    m = DiscreteModel(circuits.no_solution_circuit(), Fraction(1))
    r = emu_runner(emu_lib, m, 3)
    u = np.array([[[1.0, 1.0]], [[-1.0, -1.0]], [[1.0, 1.0]]])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = r.run(u)
    assert any("Failed to converge" in str(x.message) for x in w)
    ra = r.report_arrays()
    assert ra["n_warn"].tolist() == [0, 2, 0] and ra["first_nonconverged"].tolist() == [-1, 0, -1]
    assert np.isfinite(y).all() and y[0, 0, 0] == y[0, 0, 1]
    np.testing.assert_allclose(y[0, 0, 0], 25e-3 * np.log(1 / 1e-12 + 1), rtol=1e-9)
    r = emu_runner(emu_lib, m, 2)
    u = np.array([[[1.0, np.inf, 1.0]], [[1.0, 1.0, 1.0]]])
    with pytest.raises(AcmeError, match="non-finite"):
        r.run(u)
    ra = r.report_arrays()
    assert ra["first_nonfinite"].tolist() == [1, -1]
    y = r.run(u, check=False)          # instance 0 stays dead, instance 1 keeps running
    assert np.isnan(y[0]).all() and np.isfinite(y[1]).all()


def test_emulated_per_instance_matrices(emu_lib):
    """Monte-Carlo style batch: every instance has its own component values."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.circuit import capacitor, resistor
    from acme_jl_amd.model import DiscreteModel
    models = []
    for k in range(5):
        c = examples.diodeclipper()
        c.elements["r1"] = resistor(1e3 * (1 + 0.05 * (k - 2)))
        c.elements["c1"] = capacitor(47e-9 * (1 - 0.03 * (k - 2)))
        models.append(DiscreteModel(c, Fraction(1, 44100)))
    u = sweep_inputs("diodeclipper", 5, 150)
    from acme_jl_amd.runner import ModelRunner
    r = ModelRunner(models[0], 5, models=models, lib=emu_lib)
    y = r.run(u)
    for k in range(5):
        yref, _ = oracle_run(models[k], u[k:k + 1])
        assert_close(y[k:k + 1], yref)
    assert np.abs(y[0] - y[4]).max() > 1e-4   # the instances really differ


def test_emulated_per_instance_matrices_rare_shape(emu_lib):
    """Private model images in a RARE (kind-by-kind) shape with states: the lanes keep their rows of the
    linear update in registers (Shape::LINREG) and a block stages only the solver's part of each image."""
    from helpers import rare_per_instance_case
    from acme_jl_amd.runner import ModelRunner
    models, u = rare_per_instance_case()
    y = ModelRunner(models[0], len(models), models=models, lib=emu_lib).run(u)
    for k in range(len(models)):
        yref, _ = oracle_run(models[k], u[k:k + 1])
        assert_close(y[k:k + 1], yref)
    assert np.abs(y[0] - y[3]).max() > 1e-6   # the instances really differ


def test_emulated_monte_carlo_superover(emu_lib):
    """BASELINE config 4 in miniature (component tolerances on the fixed-pot superover): every
    instance must start from ITS model's initial solution, not the batch model's."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    from helpers import sine
    rng = np.random.Generator(np.random.PCG64(20250905))
    models = [DiscreteModel(examples.superover(1.0, 1.0, 1.0, value=lambda n, v: v * (1 + 0.05 * rng.uniform(-1, 1))),
                            Fraction(1, 44100)) for _ in range(3)]
    u = np.tile(sine(160)[None, None, :], (3, 1, 1))
    y = ModelRunner(models[0], 3, models=models, lib=emu_lib).run(u)
    for k in range(3):
        yref, _ = oracle_run(models[k], u[k:k + 1])
        assert_close(y[k:k + 1], yref, rtol=1e-9)


def test_emulated_split_run_is_bit_identical(emu_lib):
    """run! over [0,T) equals run! over [0,a) then [a,T) bit for bit: besides (x, last_p, last_z)
    the library keeps the pivot (row) order the lanes had adopted."""
    m = load("superover_fixed")
    u = sweep_inputs("superover_fixed", 4, 260)
    from acme_jl_amd.runner import ModelRunner
    y1 = ModelRunner(m, 4, lib=emu_lib).run(u)
    r = ModelRunner(m, 4, lib=emu_lib)
    y2 = np.concatenate([r.run(u[:, :, :97]), r.run(u[:, :, 97:])], axis=2)
    assert np.array_equal(y1, y2)


def test_emulated_tight_tolerance_parity(emu_lib):
    """With set_resabstol!(1e-13) on both sides stopping-test flips are harmless: the two
    implementations must agree to rounding level (helpers.RTOL_TIGHT)."""
    from helpers import RTOL_TIGHT
    from oracle.refpy import RefRunner
    m = load("birdie_var")
    u = sweep_inputs("birdie_var", 6, 300)
    r = emu_runner(emu_lib, m, 6)
    r.set_resabstol(1e-13)
    y = r.run(u)
    for i in range(6):
        rr = RefRunner(m)
        rr.set_resabstol(1e-13)
        assert_close(y[i:i + 1], rr.run(u[i])[None], rtol=RTOL_TIGHT)


def test_emulated_solver_plugin_contract(emu_lib):
    """solve / hasconverged / needediterations / extrapolation origin (src/solvers.jl:183-236,
    268-302) through acme_batch_solve, against the oracle's solver object, on solver inputs
    taken from real trajectories (random far-from-physical p make the Jacobian's condition
    number ~1e290: both Newton paths are then numerical noise)."""
    from oracle.refpy import RefRunner
    from test_gpu_parity import trajectory_ps
    rng = np.random.default_rng(3)
    for name in ("diodeclipper", "superover_fixed", "birdie_var"):
        m = load(name)
        ps = trajectory_ps(m, sweep_inputs(name, 2, 140)[1])
        N = len(ps)
        r = emu_runner(emu_lib, m, N)
        refs = [RefRunner(m) for _ in range(N)]
        for step in range(3):   # consecutive solves: the origin carries over
            p = ps[rng.permutation(N)]
            z, conv, its = r.solve(p)
            for i in range(N):
                zr, cr, ir = refs[i].solve(p[i])
                assert conv[i] == cr, (name, step, i)
                # plain Newton solves take identical iteration counts; once the homotopy
                # wrapper has to bisect, rounding-level differences change the path
                if ir <= 20:
                    assert its[i] == ir, (name, step, i, its[i], ir)
                np.testing.assert_allclose(z[i], zr, rtol=1e-7, atol=1e-10)
        x, lp, lz = r.get_state()
        for i in range(N):
            pr, zr = refs[i].get_origin(0)
            np.testing.assert_allclose(lp[i], pr, rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(lz[i], zr, rtol=1e-7, atol=1e-10)
        assert not x.any()      # solve() never touches the state vector


def test_emulated_homotopy_needed_and_failure(emu_lib):
    """A solve that only the homotopy wrapper rescues, and one with no solution
    (test/runtests.jl:207-219 in spirit, :170-183)."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd.model import DiscreteModel, SimpleSolver
    from oracle.refpy import RefRunner
    m = DiscreteModel(circuits.no_solution_circuit(), Fraction(1))
    r = emu_runner(emu_lib, m, 3)
    p = np.array([[1.0], [-1.0], [50.0]]) / m.subs[0].eq[0, 0] * m.subs[0].eq[0, 0]
    z, conv, its = r.solve(p)
    for i in range(3):
        zr, cr, ir = RefRunner(m).solve(p[i])
        assert conv[i] == cr and its[i] == ir
    ms = DiscreteModel(circuits.no_solution_circuit(), Fraction(1), solver=SimpleSolver)
    z2, conv2, its2 = emu_runner(emu_lib, ms, 3).solve(p)
    for i in range(3):
        zr, cr, ir = RefRunner(ms).solve(p[i])
        assert conv2[i] == cr and its2[i] == ir


def test_emulated_checksteady(emu_lib):
    """checksteady! (test/runtests.jl:664-671): steadystate!, tolerance 1e-13, one zero-input
    step must leave x unchanged -- for diodeclipper, birdie (fixed), superover (fixed)."""
    from acme_jl_amd.analysis import steadystate_
    for name in ("diodeclipper", "birdie_fixed", "superover_fixed"):
        m = load(name)
        r = emu_runner(emu_lib, m, 2)
        xs = steadystate_(r)
        r.set_resabstol(1e-13)
        r.run(np.zeros((2, m.nu, 1)))
        x, _, _ = r.get_state()
        np.testing.assert_allclose(x, xs, rtol=1.5e-8, atol=1e-14)
        if name != "diodeclipper":
            assert np.abs(xs).max() > 1e-4      # a real bias point (c5 charged to 9 V), not the zero state


def test_emulated_steadystate_per_instance_inputs(emu_lib):
    """N different operating points in one batched solve (per-instance derived equations)."""
    from acme_jl_amd.analysis import steadystate
    m = load("birdie_var")
    U = np.array([[0.0, 0.2], [0.0, 0.9], [0.1, 0.5]])
    X = steadystate(m, U, lib=emu_lib)
    r = emu_runner(emu_lib, m, 3)
    r.set_state(x=X)
    r.set_resabstol(1e-13)
    r.run(np.repeat(U[:, :, None], 1, axis=2))
    x, _, _ = r.get_state()
    np.testing.assert_allclose(x, X, rtol=1.5e-8, atol=1e-14)


def _simplified_superover(var, solver="HomotopySolver{SimpleSolver}"):
    """test/runtests.jl:751-756 / :782-787: superover with vb forced by an ideal source ->
    the nonlinearity decomposes into 3 (fixed pots) / 4 (pots as inputs) sub-problems."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.circuit import voltagesource
    from acme_jl_amd.model import DiscreteModel
    c = examples.superover() if var else examples.superover(1.0, 1.0, 1.0)
    c.add("vbsrc", voltagesource(4.5))
    c.connect(("vbsrc", "+"), "vb")
    c.connect(("vbsrc", "-"), "gnd")
    return DiscreteModel(c, Fraction(1 / 44100), solver)


def test_emulated_decomposed_nonlinearity(emu_lib):
    """Several sub-problems solved one after another each sample, later ones fed by the
    earlier ones through fqprev (src/ACME.jl:675-697)."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd.model import DiscreteModel
    m = _simplified_superover(False)
    assert [s.np for s in m.subs] == [2, 1, 2]
    u = sweep_inputs("superover_fixed", 5, 160)
    r = emu_runner(emu_lib, m, 5)
    assert r.kernel_shape()[:3] == (4, 12, 4)
    yref, its = oracle_run(m, u)
    assert_close(r.run(u), yref)
    assert r.report_arrays()["iters_total"].tolist() == its.tolist()
    x, p, z = r.get_state()
    assert p.shape == (5, 5) and z.shape == (5, 7)
    mv = _simplified_superover(True)
    assert [s.np for s in mv.subs] == [2, 2, 2, 4]
    uv = sweep_inputs("superover_var", 3, 100)
    yref, _ = oracle_run(mv, uv)
    assert_close(emu_runner(emu_lib, mv, 3).run(uv), yref)
    ms = DiscreteModel(circuits.series_diodes_circuit(), Fraction(1))    # test/runtests.jl:286-291
    y = emu_runner(emu_lib, ms, 1).run(np.array([[2.0], [1.0]]))
    np.testing.assert_allclose(y[:, 0], 1e-12 * (np.exp(1 / 25e-3) - 1), rtol=1.5e-8)


def test_emulated_linearize(emu_lib):
    """linearization_error! (test/runtests.jl:673-682) on the chirp the reference uses (shortened):
    birdie < 1e-7 at amplitude 1e-4 (:730), superover < 1e-4 (:749), diode clipper at 1e-3."""
    from acme_jl_amd.analysis import linearize, steadystate_
    from acme_jl_amd.runner import ModelRunner
    N = 3000
    for name, amp, bound in (("diodeclipper", 1e-3, 1e-6), ("birdie_fixed", 1e-4, 1e-7), ("superover_fixed", 1e-4, 1e-4)):
        m = load(name)
        lin = linearize(m, lib=emu_lib)
        assert not lin.subs and lin.a.shape == m.a.shape
        u = (amp * np.sin(np.pi / 2 * np.arange(N + 1) ** 2 / 50000))[None, None, :]
        r = ModelRunner(m, 1, lib=emu_lib)
        steadystate_(r)
        rl = ModelRunner(lin, 1, lib=emu_lib)
        steadystate_(rl)
        err = np.abs(r.run(u) - rl.run(u)).max()
        print(name, "linearization error", err)
        assert err < bound


def test_emulated_caching_solver(emu_lib):
    """HomotopySolver{CachingSolver{SimpleSolver}} with the GPU's bounded store (8 solutions, FIFO):
    same outputs AND the same iteration counts as the oracle's bounded variant; far fewer
    iterations than without the cache; results of the two solver stacks agree within the solver
    tolerance."""
    from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver
    from acme_jl_amd.runner import ModelRunner
    # one case per kernel family on the reference's default stack: 16-lane shared image (superover_var, birdie_var),
    # lane-per-instance kernel with a zero (diodeclipper) and a non-zero (birdie_fixed) initial solution
    for name, N, T in (("superover_var", 4, 450), ("birdie_var", 3, 600), ("diodeclipper", 3, 300), ("birdie_fixed", 4, 600)):
        m = load(name, CachingHomotopySolver)
        u = sweep_inputs(name, N, T)
        r = ModelRunner(m, N, lib=emu_lib)
        y = r.run(u)
        yref, its = oracle_run(m, u, cache_limit=16)
        assert_close(y, yref)
        # same algorithm, same Newton paths: the counts agree unless a rounding-level flip of a
        # convergence test or of a nearest-entry decision sends one side down another path
        assert np.abs(r.report_arrays()["iters_total"] - its).max() <= max(3, 0.1 * its.max())
        y2, its2 = oracle_run(m, u, solver=HomotopySolver)
        assert_close(y, y2, rtol=5e-6)    # two legal solver stacks: each within tol/g_min of the root
        if name == "superover_var":
            assert its.sum() < 0.8 * its2.sum()


def test_emulated_caching_decomposed_and_per_instance(emu_lib):
    """One solution cache per sub-problem (the reference has one CachingSolver per sub-problem),
    and per-instance model blocks start from their own initial cached point."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    m = _simplified_superover(False)
    m.solver = CachingHomotopySolver
    assert len(m.subs) == 3
    u = sweep_inputs("superover_fixed", 3, 300)
    r = ModelRunner(m, 3, lib=emu_lib)
    yref, its = oracle_run(m, u, cache_limit=16)
    assert_close(r.run(u), yref)
    assert np.abs(r.report_arrays()["iters_total"] - its).max() <= max(3, 0.1 * its.max())
    rng = np.random.Generator(np.random.PCG64(7))
    models = [DiscreteModel(examples.superover(1.0, 1.0, 1.0, value=lambda n, v: v * (1 + 0.05 * rng.uniform(-1, 1))),
                            Fraction(1, 44100), solver=CachingHomotopySolver) for _ in range(3)]
    from helpers import sine
    u = np.tile(0.7 * sine(250)[None, None, :], (3, 1, 1))
    y = ModelRunner(models[0], 3, models=models, lib=emu_lib).run(u)
    for k in range(3):
        yref, _ = oracle_run(models[k], u[k:k + 1], cache_limit=16)
        assert_close(y[k:k + 1], yref)


def test_emulated_caching_split_run_and_solve(emu_lib):
    """The solution caches persist across launches (a split run is bit-identical) and serve the
    solver plugin entry point as well."""
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    m = load("superover_fixed", CachingHomotopySolver)
    u = sweep_inputs("superover_fixed", 4, 300)
    y1 = ModelRunner(m, 4, lib=emu_lib).run(u)
    r = ModelRunner(m, 4, lib=emu_lib)
    y2 = np.concatenate([r.run(u[:, :, :131]), r.run(u[:, :, 131:])], axis=2)
    assert np.array_equal(y1, y2)


def test_emulated_sliced_host_run(emu_lib):
    """Host-buffer runs of 4096+ samples go through HBM in time slices (copies overlapped with the
    kernel on the GPU): same bits as the one-launch device path, ragged last slice included."""
    from acme_jl_amd.runner import ModelRunner
    m = load("birdie_var")
    N, T = 3, 4096 + 1000 + 7
    u = sweep_inputs("birdie_var", N, T)
    y_host = ModelRunner(m, N, lib=emu_lib).run(u)
    r = ModelRunner(m, N, lib=emu_lib)
    ub = np.ascontiguousarray(np.transpose(u, (0, 2, 1)))
    yb = np.zeros((N, T, m.ny))
    r.run_device(ub.ctypes.data, yb.ctypes.data, T)      # emulator: "device" memory is host memory
    assert np.array_equal(y_host, np.transpose(yb, (0, 2, 1)))
    yref, _ = oracle_run(m, u)
    assert_close(y_host, yref)


def test_emulated_empty_and_single_sample_runs(emu_lib):
    """Edge cases of run!: T = 0 (empty u -> empty y, state untouched), T = 1, and the 2-D
    single-instance call; pieces concatenate to the one-call result bit for bit."""
    m = load("superover_fixed")
    u = sweep_inputs("superover_fixed", 3, 40)
    r = emu_runner(emu_lib, m, 3)
    assert r.run(u[:, :, :0]).shape == (3, 1, 0)
    y = np.concatenate([r.run(u[:, :, :1]), r.run(u[:, :, 1:1]), r.run(u[:, :, 1:])], axis=2)
    yf = emu_runner(emu_lib, m, 3).run(u)
    assert np.array_equal(y, yf)
    y1 = emu_runner(emu_lib, m, 1).run(u[0])
    assert y1.shape == (1, 40) and np.array_equal(y1, yf[0])


def test_emulated_time_major_run(emu_lib):
    """run(..., time_major=True) takes (N, T, nu) / returns (N, T, ny) -- the C ABI's own layout --
    and gives the bits of the default (N, nu, T) call; shape errors are DimensionMismatch."""
    from acme_jl_amd.runner import DimensionMismatch
    m = load("superover_var")
    u = sweep_inputs("superover_var", 3, 60)
    y = emu_runner(emu_lib, m, 3).run(u)
    ut = np.ascontiguousarray(u.transpose(0, 2, 1))
    r = emu_runner(emu_lib, m, 3)
    yt = r.run(ut, time_major=True)
    assert yt.shape == (3, 60, 1) and np.array_equal(yt.transpose(0, 2, 1), y)
    out = np.empty((3, 60, 1))
    r2 = emu_runner(emu_lib, m, 3)
    assert r2.run(ut, out, time_major=True) is out and np.array_equal(out, yt)
    with pytest.raises(DimensionMismatch):
        r2.run(ut[:, :, :3], time_major=True)
    with pytest.raises(DimensionMismatch):
        r2.run(ut, np.empty((3, 59, 1)), time_major=True)


@pytest.mark.parametrize("lane_kernel", ["1", "0"])
def test_emulated_nonlinear_model_without_inputs(emu_lib, lane_kernel, monkeypatch):
    """run!(model, zeros(0, T)) on a nonlinear model with a state and NO inputs (u is NULL at the C ABI):
    both run kernels, the lane-per-instance one and the 16-lane one."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd.model import DiscreteModel
    monkeypatch.setenv("ACME_LANE_KERNEL", lane_kernel)
    m = DiscreteModel(circuits.constant_source_clipper(0.8), Fraction(1, 44100))
    assert (m.nu, m.nx) == (0, 1) and m.subs[0].nn == 2
    u = np.zeros((2, 0, 20))
    y = emu_runner(emu_lib, m, 2).run(u)
    yref, _ = oracle_run(m, u)
    assert_close(y, yref, rtol=1e-12)
    assert 0.6 < y[0, 0, -1] < 0.7          # the capacitor charges to the diode's forward voltage


@pytest.mark.parametrize("name,N,T", [("diodeclipper", 19, 200), ("birdie_fixed", 4, 200)])
def test_emulated_small_shapes_both_run_kernels(emu_lib, name, N, T, monkeypatch):
    """The shapes the lane-per-instance kernel takes by default also run -- and give the same answer to
    rounding -- in the 16-lane kernel (ACME_LANE_KERNEL=0: the A/B fallback, and the code path of
    acme_batch_solve / the Jacobian export on such batches)."""
    m = load(name)
    u = sweep_inputs(name, N, T)
    ys = {}
    for k in ("1", "0"):
        monkeypatch.setenv("ACME_LANE_KERNEL", k)
        ys[k] = emu_runner(emu_lib, m, N).run(u)
    yref, _ = oracle_run(m, u)
    assert_close(ys["1"], yref, rtol=1e-12)
    assert_close(ys["0"], yref, rtol=1e-12)


def test_emulated_multi_device_runner(emu_lib):
    """One process, several batches started asynchronously (acme_batch_run_async / acme_batch_wait), each on
    its contiguous instance range and its own slice of u / y: bit-identical to one batch of all instances.
    (The device ordinal repeats: one emulated device; on a node the ordinals are its GPUs.)"""
    from acme_jl_amd.runner import AcmeError, ModelRunner, MultiDeviceRunner
    m = load("superover_fixed")
    u = np.ascontiguousarray(sweep_inputs("superover_fixed", 7, 240).transpose(0, 2, 1))
    y1 = ModelRunner(m, 7, lib=emu_lib).run(u, time_major=True)
    mr = MultiDeviceRunner(m, 7, devices=[0, 0, 0], lib=emu_lib)
    assert mr.ranges == [(0, 3), (3, 5), (5, 7)]
    y2 = mr.run(u[:, :100])
    y2 = np.concatenate([y2, mr.run(u[:, 100:])], axis=1)       # state persists per batch
    assert np.array_equal(y1, y2)
    assert mr.report_arrays()["iters_total"].shape == (7,) and mr.get_state()[0].shape == (7, m.nx)
    # more devices than instances: empty ranges are skipped
    assert np.array_equal(MultiDeviceRunner(m, 2, devices=[0, 0, 0], lib=emu_lib).run(u[:2]), y1[:2])
    # a failing run surfaces from wait(), after every run has been joined
    bad = u.copy()
    bad[4, 3, 0] = np.inf
    mr = MultiDeviceRunner(m, 7, devices=[0, 0, 0], lib=emu_lib)
    with pytest.raises(AcmeError, match="non-finite"):
        mr.run(bad)
    assert mr.report_arrays()["first_nonfinite"].tolist() == [-1, -1, -1, -1, 3, -1, -1]


def test_emulated_low_lds_variant(emu_lib, monkeypatch):
    """The LOW-LDS kernels (model images read from HBM instead of LDS; taken when 16 private images, or the
    shared image next to the solution caches, do not fit a CU's 160 KB) compute exactly what the LDS
    kernels compute: forced on a batch that would fit (ACME_LOW_LDS=1), bit for bit; run, solve and
    Jacobian export."""
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    for name, solver in (("superover_var", None), ("superover_var", CachingHomotopySolver)):
        m = load(name, solver)
        u = sweep_inputs(name, 5, 120)
        out = {}
        for low in ("0", "1"):
            monkeypatch.setenv("ACME_LOW_LDS", low)
            r = ModelRunner(m, 5, lib=emu_lib)
            y = r.run(u)
            z, conv, its = r.solve(np.tile(r.get_state()[1][:1], (5, 1)) * 1.01)
            out[low] = (y, z, its, r.get_extrapolation_jacobian())
        for a, b in zip(out["0"], out["1"]):
            assert np.array_equal(a, b)


def test_emulated_per_instance_matrices_beyond_lds(emu_lib):
    """Monte-Carlo component tolerances on the VARIABLE-pot superover (nn = 13: 16 private images are 264 KB,
    more than a CU's LDS -- refused until round 3): every instance against the oracle run of its own,
    exactly derived model.  And the reference's default (caching) stack on a decomposed model whose four
    solution caches do not fit next to the shared image."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    from acme_jl_amd.montecarlo import derive_batch
    make = lambda value: examples.superover(value=value)     # noqa: E731   (pots as inputs)
    nominal = {}
    make(lambda name, v: nominal.setdefault(name, v))
    rng = np.random.Generator(np.random.PCG64(11))
    vals = {k: v * (1 + 0.05 * rng.uniform(-1, 1, 3)) for k, v in nominal.items()}
    batch = derive_batch(make, Fraction(1, 44100), vals, solver="HomotopySolver{SimpleSolver}")
    u = sweep_inputs("superover_var", 3, 140)
    r = ModelRunner(batch.model(0), 3, models=batch, lib=emu_lib)
    assert r.kernel_shape()[:3] == (13, 29, 11)
    y = r.run(u)
    for k in range(3):
        exact = DiscreteModel(make(lambda name, v: float(vals[name][k])), Fraction(1, 44100), "HomotopySolver{SimpleSolver}")
        yref, _ = oracle_run(exact, u[k:k + 1])
        assert_close(y[k:k + 1], yref, rtol=1e-9)
    assert np.abs(y[0] - y[1]).max() > 1e-6
    mv = _simplified_superover(True, CachingHomotopySolver)          # 4 sub-problems, medium shape
    uv = sweep_inputs("superover_var", 2, 100)
    r = ModelRunner(mv, 2, lib=emu_lib)
    assert r.kernel_shape()[:3] == (8, 24, 8)
    yref, _ = oracle_run(mv, uv, cache_limit=16)
    assert_close(r.run(uv), yref)


@pytest.mark.parametrize("name,lane", [("diodeclipper", "1"), ("diodeclipper", "0"), ("superover_fixed", "1")])
def test_emulated_whole_wave_dead(emu_lib, name, lane, monkeypatch):
    """Every instance of a wave hits a non-finite input at the same sample: from then on no lane of the wave needs a
    solve, and the solver's do-while Newton loop runs its surplus pass with every update masked -- outputs are
    NaN from that sample on, the iteration counters stop, and the samples before it are those of an undisturbed run."""
    monkeypatch.setenv("ACME_LANE_KERNEL", lane)
    m = load(name)
    N, T, K = 3, 12, 5
    u = sweep_inputs(name, N, T)
    ref = emu_runner(emu_lib, m, N)
    yref = ref.run(u[:, :, :K])
    its_ref = ref.report_arrays()["iters_total"].copy()
    ub = u.copy()
    ub[:, 0, K] = np.inf
    r = emu_runner(emu_lib, m, N)
    y = r.run(ub, check=False)
    ra = r.report_arrays()
    assert ra["first_nonfinite"].tolist() == [K] * N
    assert np.array_equal(y[:, :, :K], yref) and np.isnan(y[:, :, K:]).all()
    # the fatal sample's own iterations are counted (the solve ran before step! noticed), nothing after it
    r2 = emu_runner(emu_lib, m, N)
    r2.run(ub[:, :, :K + 1], check=False)
    assert np.array_equal(ra["iters_total"], r2.report_arrays()["iters_total"])
    assert (ra["iters_total"] >= its_ref).all()


def test_emulated_lane_kernel_cache_layout(emu_lib, monkeypatch):
    """The lane-per-instance kernel and the 16-lane kernels share ONE HBM layout of the solution cache
    (cp | count, head | cz): after a lane-kernel run that stored solutions, solve(p = 0) through the 16-lane
    solve kernel must hit the CachingSolver's initial entry (p = 0, z = init_z; src/solvers.jl:327-333) and
    accept it as it stands -- one iteration, z == init_z.  (Round 3's lane kernel wrote its stored z's two
    doubles low: entry 1 landed on entry 0.)"""
    from acme_jl_amd.model import CachingHomotopySolver
    monkeypatch.setenv("ACME_LANE_KERNEL", "1")
    m = load("birdie_fixed", CachingHomotopySolver)
    assert np.abs(m.subs[0].init_z).min() > 0.01
    u = sweep_inputs("birdie_fixed", 4, 300)
    r = emu_runner(emu_lib, m, 4)
    r.run(u[:, :, :170])
    r.run(u[:, :, 170:])
    assert (r.report_arrays()["iters_max"] > 5).sum() >= 2      # something was stored
    z, conv, its = r.solve(np.zeros((4, 2)))
    assert conv.all() and its.tolist() == [1, 1, 1, 1]
    assert np.array_equal(z, np.tile(m.subs[0].init_z, (4, 1)))


def test_emulated_condensed_kernel_is_selected_and_exact(emu_lib, monkeypatch):
    """The headline model runs in the CONDENSED kernel (its 6 potentiometer rows eliminated once per change of the
    pot positions, Newton on the other 7) and walks the oracle's Newton path: identical iteration totals, outputs
    to rounding -- with fixed pots, with pots that move every sample (incl. the reference's own test input,
    test/runtests.jl:778, whose first sample is the singular drive = 1.0 corner), on both solver stacks; the plain
    13 x 13 kernel (ACME_CONDENSE=0) still does the same."""
    from acme_jl_amd.model import CachingHomotopySolver
    from helpers import HS, RTOL_SAME, moving_pot_inputs
    for solver, lim in ((HS, None), (CachingHomotopySolver, 16)):
        m = load("superover_var", solver)
        for u in (sweep_inputs("superover_var", 5, 160, seed=2), moving_pot_inputs(4, 220)):
            yref, its = oracle_run(m, u, cache_limit=lim)
            for cond in ("1", "0"):
                monkeypatch.setenv("ACME_CONDENSE", cond)
                r = emu_runner(emu_lib, m, u.shape[0])
                assert r.kernel_variant() == ((6, False) if cond == "1" else (0, False))
                y = r.run(u, check=False)
                assert_close(y, yref, rtol=RTOL_SAME)
                assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (solver, cond)


def test_emulated_condensed_state_solve_and_jacobian(emu_lib, monkeypatch):
    """The condensed kernel keeps z in a permuted basis internally: state get / set, acme_batch_solve and the
    extrapolation Jacobian speak the caller's; a run split at a launch boundary -- with the state read and written
    back in between -- repeats the one-launch run bit for bit, and the condensed and plain kernels agree."""
    from helpers import HS
    m = load("superover_var", HS)
    u = sweep_inputs("superover_var", 3, 120, seed=4)
    out = {}
    for cond in ("1", "0"):
        monkeypatch.setenv("ACME_CONDENSE", cond)
        y1 = emu_runner(emu_lib, m, 3).run(u)
        r = emu_runner(emu_lib, m, 3)
        ya = r.run(u[:, :, :50])
        yb = r.run(u[:, :, 50:])
        assert np.array_equal(y1, np.concatenate([ya, yb], axis=2))
        x, p, z = r.get_state()
        jac = r.get_extrapolation_jacobian()
        zs, conv, its = r.solve(p * (1 + 1e-9))       # (next to the origin: a plain Newton solve, no homotopy episode)
        out[cond] = (y1, x, p, z, jac, zs, conv, its)
        # a state written back from outside is not trusted to satisfy the linear rows: same results to rounding
        r2 = emu_runner(emu_lib, m, 3)
        r2.run(u[:, :, :50])
        x2, p2, z2 = r2.get_state()
        r2.set_state(x=x2, p=p2, z=z2)
        yb2 = r2.run(u[:, :, 50:])
        assert_close(yb2, yb, rtol=1e-12)
    names = ("y", "x", "p", "z", "jac", "z of solve", "converged", "iterations")
    for name, a, b in zip(names, out["1"], out["0"]):      # (small entries of the Jacobian are differences of large ones)
        b = np.asarray(b, dtype=float)
        # the solve from an arbitrary p stops at the residual tolerance: two eliminations, two last iterates
        rtol = 1e-6 if name == "z of solve" else 1e-9
        np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=rtol, atol=1e-12 + 1e-12 * np.abs(b).max(), err_msg=name)


def test_emulated_progress_callback_and_host_buffer_release(emu_lib):
    """@showprogress of run!(runner, y, u) (src/ACME.jl:587-604,653) as a callback after every time slice of a
    host-buffer run, and the page-locked caller arrays being released on demand (the emulator backend only counts
    the ranges; the bookkeeping is the library's)."""
    m = load("diodeclipper")
    seen = []
    from acme_jl_amd.runner import ModelRunner
    r = ModelRunner(m, 3, lib=emu_lib, showprogress=lambda done, total: seen.append((done, total)))
    T = 4500                                   # >= 4096: sliced
    u = sweep_inputs("diodeclipper", 3, T)
    y = r.run(u)
    assert seen and seen[-1] == (T, T)
    dones = [d for d, _ in seen]
    assert dones == sorted(dones) and len(seen) >= 2 and all(t == T for _, t in seen)
    yref, _ = oracle_run(m, u)
    assert_close(y, yref, rtol=1e-12)
    seen.clear()
    r.run(u[:, :, :100])                       # a short run reports once, at its end
    assert seen == [(100, 100)]
    r.release_host_buffers()
    r.release_host_buffers()                   # idempotent


def test_emulated_host_buffer_pipelines_agree(emu_lib, monkeypatch):
    """run! through host buffers with the 16-lane kernel: the default, streamed pipeline (one launch, u copied into HBM
    chunk by chunk -- KArgs::u_ready; the emulator launches synchronously, so here the copy comes first -- y written in
    place), the sliced one (ACME_HOST_SLICES: y in place, the first time slice of u read in place, the following ones
    staged under the kernel of the slice before; KArgs::u_stride / y_stride), the whole run in place
    (ACME_HOST_SLICES=1) and the fully staged pipeline (ACME_HOST_ZEROCOPY=0) give the same bits, the sliced ones
    report progress slice by slice, and the result is the oracle's."""
    monkeypatch.setenv("ACME_LANE_KERNEL", "0")
    monkeypatch.setenv("ACME_HOST_REGISTER_MIN", "1024")        # (the library page-locks arrays of 1 MB and more)
    m = load("diodeclipper")
    from acme_jl_amd.runner import ModelRunner
    N, T = 3, 4500
    u = sweep_inputs("diodeclipper", N, T)
    out = {}
    for name, env in (("default", {}), ("8 slices", {"ACME_HOST_SLICES": "8"}), ("in place", {"ACME_HOST_SLICES": "1"}),
                      ("staged", {"ACME_HOST_ZEROCOPY": "0"}), ("3 slices", {"ACME_HOST_SLICES": "3"}),
                      ("not streamed", {"ACME_HOST_STREAM": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        seen = []
        # (a progress callback asks for time slices: the streamed pipeline is one launch)
        r = ModelRunner(m, N, lib=emu_lib, showprogress=False if name == "default" else lambda done, total: seen.append((done, total)))
        r.set_host_retention(True)                 # (these pipelines work on page-locked, mapped arrays: the runner holds them)
        y1 = r.run(u)
        y2 = r.run(u[:, :, :4200])                 # (a second call on a sub-range of the same arrays' shape)
        out[name] = (y1, y2, [d for d, _ in seen])
        for k in env:
            monkeypatch.delenv(k)
    for name in ("8 slices", "in place", "staged", "3 slices", "not streamed"):
        assert np.array_equal(out[name][0], out["default"][0]) and np.array_equal(out[name][1], out["default"][1]), name
    assert out["8 slices"][2][:5] == [1024, 2048, 3072, 4096, 4500]         # 8 wanted, 1 024 samples at least: 5 slices
    assert out["not streamed"][2][:5] == [1024, 2048, 3072, 4096, 4500]
    assert out["in place"][2][:1] == [4500]
    assert out["3 slices"][2][:3] == [1504, 3008, 4500]
    yref, _ = oracle_run(m, u)
    assert_close(out["default"][0], yref, rtol=1e-12)
    # the default: nothing of the caller's arrays is kept (acme_batch_set_host_retention off) -- the staged pipeline from
    # ordinary memory, same bits; no range stays "page-locked" behind the call
    r = ModelRunner(m, N, lib=emu_lib)
    y1 = r.run(u)
    assert np.array_equal(y1, out["default"][0])
    assert getattr(r, "_held", None) is None


def test_emulated_generic_kernel_never_refuses(emu_lib, monkeypatch):
    """Models beyond every tuned kernel shape (20 unknowns, 9 sub-problems, 40 states) run in the generic
    lane-per-instance kernel and walk the oracle's path (identical iteration totals, outputs to rounding), on both
    solver stacks and across a launch boundary; the same kernel, forced onto the BASELINE models (ACME_GENERIC=1),
    agrees with the oracle as well, and its solve / Jacobian / state entry points with the tuned kernels'."""
    from acme_jl_amd.model import CachingHomotopySolver
    from helpers import HS, RTOL_SAME, beyond_the_tuned_shapes
    for name, m, u in beyond_the_tuned_shapes():
        assert max([s.nn for s in m.subs] + [0]) > 16 or len(m.subs) > 8 or m.nx > 32, name
        for solver, lim in ((HS, None), (CachingHomotopySolver, 16)):
            m.solver = solver
            r = emu_runner(emu_lib, m, u.shape[0])
            assert r.kernel_variant() == (0, True), name
            y = np.concatenate([r.run(u[:, :, :70]), r.run(u[:, :, 70:])], axis=2)
            yref, its = oracle_run(m, u, cache_limit=lim)
            assert_close(y, yref, rtol=RTOL_SAME)
            assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (name, solver)
    monkeypatch.setenv("ACME_GENERIC", "1")
    for name, N, T in (("diodeclipper", 3, 150), ("superover_var", 2, 100)):
        for solver, lim in ((HS, None), (CachingHomotopySolver, 16)):
            m = load(name, solver)
            u = sweep_inputs(name, N, T)
            r = emu_runner(emu_lib, m, N)
            yref, its = oracle_run(m, u, cache_limit=lim)
            assert_close(r.run(u), yref, rtol=RTOL_SAME)
            assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (name, solver)
    m = load("superover_fixed", HS)
    u = sweep_inputs("superover_fixed", 3, 80)
    got = {}
    for gen in ("1", "0"):
        monkeypatch.setenv("ACME_GENERIC", gen)
        r = emu_runner(emu_lib, m, 3)
        r.run(u)
        x, p, z = r.get_state()
        jac = r.get_extrapolation_jacobian()
        zs, conv, its = r.solve(p * (1 + 1e-9))
        r.set_state(x=x, p=p, z=z)
        y2 = r.run(u[:, :, :20])
        got[gen] = (x, p, z, jac, zs, conv, its, y2)
    for a, b in zip(got["1"], got["0"]):
        b = np.asarray(b, dtype=float)
        np.testing.assert_allclose(np.asarray(a, dtype=float), b, rtol=1e-9, atol=1e-12 + 1e-12 * np.abs(b).max())


def test_emulated_up_to_eight_sub_problems(emu_lib):
    """Five to eight nonlinear sub-problems (src/ACME.jl:675-697 loops over however many nldecompose! left) run on a tuned
    16-lane shape -- not in the lane-per-instance generic kernel, as up to round 5: the oracle's outputs and iteration
    totals on both solver stacks, across a launch boundary."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
    from helpers import HS, RTOL_SAME, sine
    u = np.array([0.2, 1.0, 3.0])[:, None, None] * sine(160)[None, None, :]
    for stages in (5, 8):
        for solver, lim in ((HS, None), (CachingHomotopySolver, 16)):
            m = DiscreteModel(circuits.buffered_clipper_chain(stages), Fraction(1, 44100), solver)
            assert len(m.subs) == stages
            r = emu_runner(emu_lib, m, u.shape[0])
            assert r.kernel_family() == "tuned", stages
            y = np.concatenate([r.run(u[:, :, :70]), r.run(u[:, :, 70:])], axis=2)
            yref, its = oracle_run(m, u, cache_limit=lim)
            assert_close(y, yref, rtol=RTOL_SAME)
            assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (stages, solver)


def test_emulated_isolation_of_slow_instances(emu_lib):
    """acme_batch_set_isolation: the instances that needed more than a threshold of iterations per sample over the
    previous run get a launch of their own (KArgs::inst_map), the others theirs -- and nothing anybody computes
    changes, bit for bit; groups are re-formed run after run.  (Threshold set to the sweep's median here, so that the
    batch really splits; the GPU test runs the singular drive = 1.0 cells this is meant for.)"""
    from helpers import HS
    m = load("superover_var", HS)
    N, T = 9, 60
    u = sweep_inputs("superover_var", N, T, seed=11)
    ref = emu_runner(emu_lib, m, N)
    y_ref = [ref.run(u) for _ in range(3)]
    per = ref.report_arrays()["iters_total"] / (3 * T)
    thr = float(np.median(per))
    assert per.min() < thr < per.max()
    r = emu_runner(emu_lib, m, N)
    r.set_isolation(thr)
    y = [r.run(u) for _ in range(3)]      # run 1: one launch (nothing known yet); runs 2 and 3: two groups
    for a, b in zip(y, y_ref):
        assert np.array_equal(a, b)
    ra, rb = r.report_arrays(), ref.report_arrays()
    for k in ("iters_total", "n_warn", "iters_max", "first_nonconverged"):
        assert np.array_equal(ra[k], rb[k]), k
    for a, b in zip(r.get_state(), ref.get_state()):
        assert np.array_equal(a, b)
    z, conv, its = r.solve(r.get_state()[1])          # (entry points that read the batch see it complete)
    assert conv.all()
    r.set_isolation(0.0)
    assert np.array_equal(r.run(u), ref.run(u))


def test_emulated_balance_is_invisible(emu_lib, monkeypatch):
    """acme_batch_set_balance: the waves (groups of 4 consecutive instances) are dealt to the launch's slots by the
    Newton iterations they needed since the last placement -- heaviest first, the lightest on top of the heaviest when
    the launch has two rounds of blocks -- and nothing anybody computes changes, bit for bit; the slots an incomplete
    last wave leaves are empty.  (A "chip" of one compute unit, so that 26 instances are two rounds.)"""
    from helpers import HS
    monkeypatch.setenv("ACME_EMU_CUS", "1")
    monkeypatch.setenv("ACME_BALANCE_MIN_SAMPLES", "20")
    m = load("superover_var", HS)
    N, T = 26, 30
    u = sweep_inputs("superover_var", N, T, seed=5)
    ref = emu_runner(emu_lib, m, N).set_balance(0)
    r = emu_runner(emu_lib, m, N).set_balance(1)
    assert np.array_equal(r.placement(), np.arange(N))
    places = []
    for k in range(3):
        assert np.array_equal(r.run(u), ref.run(u)), k
        places.append(r.placement())
    assert np.array_equal(places[0], np.arange(N))             # the first launch had nothing to go by
    assert np.array_equal(ref.placement(), np.arange(N))
    p = places[1]                                              # 7 waves, 28 slots
    assert len(p) == 28 and sorted(p[p >= 0].tolist()) == list(range(N)) and (p < 0).sum() == 2
    for q in range(7):
        w = p[4 * q:4 * q + 4]
        assert w[0] % 4 == 0 and all(w[j] == (w[0] + j if w[0] + j < N else -1) for j in range(4))
    ra, rb = r.report_arrays(), ref.report_arrays()
    for key in ("iters_total", "n_warn", "iters_max", "first_nonconverged"):
        assert np.array_equal(ra[key], rb[key]), key
    for a, b in zip(r.get_state(), ref.get_state()):
        assert np.array_equal(a, b)
    # first launch's weights -> second launch's slots: 4 heaviest in rank order, then the three lightest, lightest first
    r2 = emu_runner(emu_lib, m, N).set_balance(1)
    r2.run(u)
    it = np.concatenate([r2.report_arrays()["iters_total"], [0, 0]])
    w = it.reshape(7, 4).max(axis=1)
    r2.run(u)
    order = sorted(range(7), key=lambda k: (-w[k], k))
    want = order[:4] + order[4:][::-1]
    assert (r2.placement()[::4] // 4).tolist() == want


def test_emulated_empty_slots_in_the_placement(emu_lib, monkeypatch):
    """Slots of a launch may be empty (inst_map = -1).  The library leaves only those of an incomplete last wave empty;
    the developer's knob ACME_WAVE_DENSITY fills waves with two or one instance instead of four, which exercises them
    everywhere: across launches, sub-problems and the state entry points, bit-identical to full waves."""
    from helpers import HS
    from acme_jl_amd.model import CachingHomotopySolver
    for name, solver, N, per in (("superover_var", CachingHomotopySolver, 5, 1), ("birdie_var", HS, 11, 2)):
        m = load(name, solver)
        T = 50
        u = sweep_inputs(name, N, T, seed=3)
        monkeypatch.delenv("ACME_WAVE_DENSITY", raising=False)
        ref = emu_runner(emu_lib, m, N)
        y_ref = [ref.run(u), ref.run(u)]
        assert np.array_equal(ref.placement(), np.arange(N))
        monkeypatch.setenv("ACME_WAVE_DENSITY", str(per))
        r = emu_runner(emu_lib, m, N)
        y = [r.run(u), r.run(u)]
        p = r.placement()
        nu = -(-N // per)
        assert len(p) == 4 * nu and sorted(p[p >= 0].tolist()) == list(range(N)), (name, p)
        assert all((p[4 * q:4 * q + 4] >= 0).sum() == min(per, N - per * (p[4 * q] // per)) for q in range(nu)), (name, p)
        for a, b in zip(y, y_ref):
            assert np.array_equal(a, b), name
        ra, rb = r.report_arrays(), ref.report_arrays()
        for key in ("iters_total", "n_warn", "iters_max"):
            assert np.array_equal(ra[key], rb[key]), (name, key)
        for a, b in zip(r.get_state(), ref.get_state()):
            assert np.array_equal(a, b), name


def test_emulated_mosfet_polynomial_cap_is_reported_by_the_abi(emu_lib):
    """The reference's MOSFET takes threshold / gain polynomials of any length (src/elements.jl:436-450); the element
    table holds 4 coefficients each -- more is refused by acme_model_add_subproblem itself, with a message."""
    import ctypes as C
    L = emu_lib.L
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    mh = C.c_void_p()
    z = (C.c_double * 1)(0.0)
    emu_lib.check(L.acme_model_create(0, 0, 0, 1, z, z, z, z, z, z, z, z, C.byref(mh)))
    par = np.zeros(16)
    par[:12] = [1.0, 0.0, 5.0, 0.1, 0.2, 0.3, 0.4, 1.0, 2.0, 0.0, 0.0, 0.0]      # nvt = 5
    one = np.ones(3)
    kind, qoff, roff = (np.array([v], dtype=np.int32) for v in (4, 0, 0))
    rc = L.acme_model_add_subproblem(mh, 1, 3, 1, one.ctypes.data_as(dp), one.ctypes.data_as(dp), one.ctypes.data_as(dp),
                                     one.ctypes.data_as(dp), one.ctypes.data_as(dp), one.ctypes.data_as(dp),
                                     one.ctypes.data_as(dp), 1, kind.ctypes.data_as(ip), qoff.ctypes.data_as(ip),
                                     roff.ctypes.data_as(ip), par.ctypes.data_as(dp))
    assert rc < 0 and b"1 ... 4 coefficients" in L.acme_last_error()
    L.acme_model_destroy(mh)


@pytest.mark.timeout(900)          # (a kernel that stops converging runs the emulator to its iteration limits: fail, do not crawl)
def test_emulated_mid_size_kernel(emu_lib, monkeypatch):
    """The cooperative mid-size kernel (csrc/acme_coop.h: one instance per 16 lanes, rows dealt out over the lanes, working
    arrays in LDS) on one sub-problem of 20 / 24 / 32 unknowns: the oracle's outputs and iteration totals on both solver
    stacks, across a launch boundary; the same from private model images (the image then stays in HBM); and the
    lane-per-instance kernel (ACME_COOP=0) agrees -- the two restate the same arithmetic entry by entry."""
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    from helpers import HS, RTOL_SAME, beyond_the_tuned_shapes, mid_size_models
    for name, m, u in mid_size_models(more=True) + beyond_the_tuned_shapes()[:1]:
        for solver, lim in ((HS, None), (CachingHomotopySolver, 16)):
            m.solver = solver
            yref, its = oracle_run(m, u, cache_limit=lim)
            outs = {}
            # the threshold path on a matrix in LDS (what 33 ... 64 unknowns run: one instance per wave, one row per lane;
            # ACME_COOP_REG=0 sends 17 ... 32 unknowns there too, with 16 lanes per instance; ACME_COOP_WAVE64 = 1 / 0 pins the
            # lanes per instance), from the natural row order and from the reversed one -- which the first elimination
            # cannot keep: the re-learning path (the reference's pivoting on the matrix in LDS) runs at once
            envs = [{"ACME_COOP_REG": "0", "ACME_COOP_WAVE64": "1"}, {"ACME_COOP_REG": "0", "ACME_COOP_WAVE64": "0"}]
            if name in ("27 unknowns", "34 unknowns", "20 unknowns"):          # (every variant on an odd size, on the first size beyond the registers, on the bench's)
                envs += [{"ACME_COOP_REG": "0"}, {"ACME_COOP_REG": "0", "ACME_COOP_ORDER": "reversed"},
                         {"ACME_COOP_REG": "0", "ACME_COOP_WAVE64": "0", "ACME_COOP_ORDER": "reversed"}]
            for env in envs:
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                r = ModelRunner(m, u.shape[0], lib=emu_lib)
                assert r.kernel_family() == "coop", (name, env)
                y = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
                assert_close(y, yref, rtol=RTOL_SAME)
                assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (name, solver, env)
                key = "lds64" if env.get("ACME_COOP_WAVE64") == "1" else "lds16" if env.get("ACME_COOP_WAVE64") == "0" else "lds"
                if "ACME_COOP_ORDER" not in env:
                    outs[key] = y
                for k in env:
                    monkeypatch.delenv(k)
            # (16 or 64 lanes per instance: who computes a row differs, not what is computed)
            assert np.array_equal(outs["lds16"], outs["lds64"]), name
            for variant in ("coop", "coop, private images", "coop, literal", "lane per instance"):
                if variant == "lane per instance":
                    monkeypatch.setenv("ACME_COOP", "0")
                else:
                    monkeypatch.delenv("ACME_COOP", raising=False)
                # (17 ... 32 unknowns run the instantiations with the Jacobian's rows in registers: elimination in a learnt
                # row order, |l| <= 8; 33 ... 64 the same scheme on a matrix in LDS; ACME_COOP_LITERAL=1 selects the any-size
                # instantiation -- the reference's pivoting, factors in LDS)
                if "literal" in variant:
                    monkeypatch.setenv("ACME_COOP_LITERAL", "1")
                else:
                    monkeypatch.delenv("ACME_COOP_LITERAL", raising=False)
                models = [m] * u.shape[0] if "private" in variant else None
                r = ModelRunner(m, u.shape[0], lib=emu_lib, models=models)
                assert r.kernel_family() == ("generic" if variant == "lane per instance" else "coop"), (name, variant)
                y = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
                assert_close(y, yref, rtol=RTOL_SAME)
                assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (name, solver, variant)
                outs[variant] = y
            monkeypatch.delenv("ACME_COOP", raising=False)
            assert np.array_equal(outs["coop"], outs["coop, private images"]), name
            # the launch shape (waves per block sharing the staged tables / image, instances per wave, image in LDS or
            # not: csrc/acme_api.inc coop_shape) does not show in the bits
            if solver is HS:
                for lit, wpb, gpw, imgl in (("0", "4", "4", "1"), ("0", "3", "2", "0"), ("0", "1", "1", "1"),
                                            ("1", "3", "2", "0"), ("1", "2", "1", "1"), ("1", "1", "4", "0")):
                    for k, v in (("ACME_COOP_LITERAL", lit), ("ACME_COOP_WPB", wpb), ("ACME_COOP_GPW", gpw), ("ACME_COOP_IMGL", imgl)):
                        monkeypatch.setenv(k, v)
                    r = ModelRunner(m, u.shape[0], lib=emu_lib)
                    y = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
                    assert np.array_equal(y, outs["coop, literal" if lit == "1" else "coop"]), (name, lit, wpb, gpw, imgl)
                    assert r.report_arrays()["iters_total"].tolist() == its.tolist(), (name, lit, wpb, gpw, imgl)
                for k in ("ACME_COOP_LITERAL", "ACME_COOP_WPB", "ACME_COOP_GPW", "ACME_COOP_IMGL"):
                    monkeypatch.delenv(k)
            assert np.abs(outs["coop, literal"] - outs["lane per instance"]).max() <= 1e-13 * max(1.0, np.abs(yref).max()), name
            assert np.abs(outs["coop"] - outs["coop, literal"]).max() <= RTOL_SAME * max(1.0, np.abs(yref).max()), name


def test_emulated_birdie_var_iteration_gap_is_rounding(emu_lib, monkeypatch):
    """tests/birdie_gap.py: where and why the kernels' iteration totals part from the oracle's on the birdie_var sweep
    (VERDICT r5): instances 14 ... 17 only; first at the samples of birdie_gap.PARTING; there the reference's direct attempt
    overflows after 3 iterations (homotopy: 35 in all) while the kernel's converges by itself in 51, both reaching the same z;
    the lane-per-instance generic kernel, which runs the reference's LU literally, parts from the oracle as well (so the
    16-lane kernels' threshold pivoting is NOT the cause); and the oracle's own counts move with 1-ulp changes of p."""
    import birdie_gap as bg
    m = load(bg.NAME)
    u = sweep_inputs(bg.NAME, bg.N, bg.T)
    yref, its = oracle_run(m, u)

    def make(n, model=None):
        return emu_runner(emu_lib, model or m, n)

    totals = {}
    for kernel in ("16-lane", "lane per instance, literal LU"):
        if kernel != "16-lane":
            monkeypatch.setenv("ACME_GENERIC", "1")
        r = make(bg.N)
        y = r.run(u)
        assert_close(y, yref)
        totals[kernel] = r.report_arrays()["iters_total"]
        monkeypatch.delenv("ACME_GENERIC", raising=False)
        d = totals[kernel] - its
        assert (d[:14] == 0).all() and (d[14:] != 0).all(), (kernel, d.tolist())
    print("birdie_var iteration totals: oracle", int(its.sum()), {k: int(v.sum()) for k, v in totals.items()})
    assert int(its.sum()) == 95012 and int(totals["16-lane"].sum()) == 102981
    assert not np.array_equal(totals["16-lane"], totals["lane per instance, literal LU"])      # (each kernel its own path)
    for inst in (14, 17):
        io, ik = bg.per_sample_iterations(make, m, u[inst], bg.PARTING[inst] + 1)
        first = int(np.argmax(io != ik))
        assert (io != ik).any() and first == bg.PARTING[inst], (inst, first)
    state, out = bg.direct_attempts(make, m, u[17], 2)
    (oc, oi), (kc, ki), _ = out["SimpleSolver"]
    assert (oc, oi) == (False, 3) and (kc, ki) == (True, 51), out      # the reference's direct attempt fails, the kernel's does not
    (oc, oi), (kc, ki), dz = out["HomotopySolver{SimpleSolver}"]
    assert oc and kc and (oi, ki) == (35, 51) and dz < 1e-10, out      # ... and both solves end at the same z
    # the oracle against ITSELF: p[0] moved by up to four ulps
    rows = bg.oracle_under_ulps(m, state)
    assert len({r[3] for r in rows}) >= 3, rows                         # the full solve's iteration count is not stable
    rows38 = bg.oracle_under_ulps(m, bg.direct_attempts(make, m, u[17], 38)[0])
    assert {r[1] for r in rows38} == {True, False}, rows38              # whether the direct attempt converges is not either
    print("oracle at (17, 2) under ulps of p[0]:", rows)
    print("oracle at (17, 38) under ulps of p[0]:", rows38)


def test_emulated_constant_input_rows(emu_lib):
    """acme_batch_run_const: input rows that keep one value per instance for the whole call are handed
    over once, the others as [N][T][nu_var]; the library puts the full rows together on the device.  The results are those
    of run on the materialised input, bit for bit -- on the headline model (three potentiometer rows of four inputs; long
    enough for several time slices), with another choice of rows, with every row constant and with none; wrong shapes are
    refused."""
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import DimensionMismatch
    m = load("superover_var", CachingHomotopySolver)
    N, T = 3, 4200                                   # (4 096 samples and more run in time slices)
    u = sweep_inputs("superover_var", N, T, seed=3)  # [N][nu][T]: row 0 the signal, rows 1 .. 3 the pots
    ub = np.ascontiguousarray(u.transpose(0, 2, 1))
    y_ref = emu_runner(emu_lib, m, N).run(ub, time_major=True)
    r = emu_runner(emu_lib, m, N)
    y = r.run_const(ub[:, :, :1], ub[:, 0, :], (1, 2, 3))
    assert np.array_equal(y, y_ref)
    # rows 1 and 3 constant, 0 and 2 varying (in row order)
    r = emu_runner(emu_lib, m, N)
    assert np.array_equal(r.run_const(ub[:, :300, (0, 2)], ub[:, 0, :], (3, 1)), y_ref[:, :300])
    # none constant = run; all constant = a constant input
    r = emu_runner(emu_lib, m, N)
    assert np.array_equal(r.run_const(ub[:, :300], ub[:, 0, :], ()), y_ref[:, :300])
    uc = ub[:, :200].copy()
    uc[:, :, 0] = 0.3
    r = emu_runner(emu_lib, m, N)
    assert np.array_equal(r.run_const(np.zeros((N, 200, 0)), uc[:, 0, :], (0, 1, 2, 3)), emu_runner(emu_lib, m, N).run(uc, time_major=True))
    with pytest.raises(DimensionMismatch):
        emu_runner(emu_lib, m, N).run_const(ub[:, :, :2], ub[:, 0, :], (1, 2, 3))
    with pytest.raises(DimensionMismatch):
        emu_runner(emu_lib, m, N).run_const(ub[:, :, :1], ub[:, 0, :3], (1, 2, 3))


def element_parameter_sweeps():
    """(name, models, u[N, nu, T]): batches whose instances differ in ELEMENT parameters -- every model of the reference
    carries its own element closures (src/elements.jl:236-245, 309-406): a diode clipper swept over the diodes' saturation
    currents and emission coefficients, and transistors with different Gummel-Poon refinements switched on (different
    compile-time branches of the reference's closure, :331-396) in one batch."""
    from fractions import Fraction
    import circuits
    from acme_jl_amd import examples
    from acme_jl_amd.model import DiscreteModel
    from helpers import HS, sine
    t = Fraction(1, 44100)
    n = 20
    clip = [DiscreteModel(examples.diodeclipper(is1=1e-15 * 10 ** (3 * k / (n - 1)), is2=1.8e-15 * 10 ** (2 * (n - 1 - k) / (n - 1)),
                                                eta1=1 + 0.05 * (k % 3), eta2=1 + 0.04 * (k % 4)), t, HS) for k in range(n)]
    u_clip = np.linspace(0.2, 3.0, n)[:, None, None] * sine(150)[None, None, :]
    isc, ise, etac, etae, bf, br = 1e-6, 2e-6, 1.1, 1.0, 100, 10
    gp = []
    for bits in (0, 1, 6, 17, 40, 64, 129, 200, 255):
        kw = dict(ile=50e-9 if bits & 1 else 0, ilc=100e-9 if bits & 2 else 0, etacl=1.2 if bits & 4 else etac,
                  etael=1.1 if bits & 8 else etae, vaf=10 if bits & 16 else np.inf, var=50 if bits & 32 else np.inf,
                  ikf=50e-3 if bits & 64 else np.inf, ikr=500e-3 if bits & 128 else np.inf)
        gp.append(DiscreteModel(circuits.bjt_test_circuit("npn", isc=isc * (1 + 0.1 * len(gp)), ise=ise, etac=etac, etae=etae, bf=bf + 10 * len(gp), br=br, **kw),
                                Fraction(1), HS))
    u_gp = np.tile(circuits.bjt_test_input("npn")[None], (len(gp), 1, 1))
    return [("diode clipper, per-instance is / eta", clip, u_clip), ("BJT test circuit, per-instance Gummel-Poon branches", gp, u_gp)]


def test_emulated_per_instance_element_parameters(emu_lib):
    """acme_batch_set_matrices with models that differ in element PARAMETERS (same circuit structure): every instance
    gets its own element table, a block stages its 16 tables in LDS -- against oracle runs of each instance's own model
    (identical iteration totals); models of ANOTHER structure are refused."""
    from acme_jl_amd import examples
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import AcmeError, ModelRunner
    from fractions import Fraction
    from helpers import HS, RTOL_SAME
    for name, models, u in element_parameter_sweeps():
        r = ModelRunner(models[0], len(models), lib=emu_lib, models=models)
        y = np.concatenate([r.run(u[:, :, :40]), r.run(u[:, :, 40:])], axis=2)
        its = r.report_arrays()["iters_total"]
        for k, m in enumerate(models):
            yref, iref = oracle_run(m, u[k:k + 1])
            assert_close(y[k:k + 1], yref, rtol=RTOL_SAME)
            assert its[k] == iref[0], (name, k)
        assert np.abs(y[0] - y[-1]).max() > 1e-6, name        # (the parameters matter)
    with pytest.raises(AcmeError):
        ModelRunner(models[0], 2, lib=emu_lib, models=[models[0], DiscreteModel(examples.diodeclipper(), Fraction(1, 44100), HS)])


def superover_models_with_their_own_diodes(n, solver):
    """n superover models (pots as inputs) that differ in their diodes' saturation currents only"""
    import copy
    base = load("superover_var", solver)
    models = []
    for k in range(n):
        m = copy.deepcopy(base)
        for e in m.subs[0].table:
            if e["kind"] == 1:                     # the diodes: another saturation current per instance
                e["par"] = [e["par"][0] * (1.0 + 0.5 * k), e["par"][1]]
        models.append(m)
    return models


def test_emulated_per_instance_elements_and_condensed_shapes(emu_lib, monkeypatch):
    """The condensed kernel shapes (potentiometers as inputs) do not carry the per-instance element path (it cost the
    headline kernel 0.8 % by its mere presence).  A batch of superover models with their own diode parameters moves to the
    plain 13 x 13 shape BY ITSELF (the library chooses, not the environment) -- whether the differing
    models come with the first acme_batch_set_matrices call, with a later one (the instances set before keep their models),
    or after the batch has run (every instance keeps its state) -- and matches each model's own oracle run;
    acme_batch_kernel_variant reports the move."""
    from acme_jl_amd.runner import ModelRunner
    from helpers import HS, RTOL_SAME
    models = superover_models_with_their_own_diodes(3, HS)
    u = sweep_inputs("superover_var", 3, 60, seed=2)
    ref = [oracle_run(m, u[k:k + 1]) for k, m in enumerate(models)]

    def check(r, y, ks=range(3)):
        for k in ks:
            assert_close(y[k:k + 1], ref[k][0], rtol=RTOL_SAME)
            assert r.report_arrays()["iters_total"][k] == ref[k][1][0], k
    # all at once
    r = ModelRunner(models[0], 3, lib=emu_lib, models=models)
    assert r.kernel_variant()[0] > 0 and r.batch_kernel_variant() == (0, "tuned")      # the MODEL condenses, the batch has moved
    y = r.run(u)
    check(r, y)
    assert np.abs(y[0] - y[2]).max() > 1e-9
    # in two calls: equal models first (the batch stays condensed), then the ones that differ
    r2 = ModelRunner(models[0], 3, lib=emu_lib, models=[models[0]] * 3)
    assert r2.batch_kernel_variant()[0] > 0
    r2.set_models(2, [models[2]])
    assert r2.batch_kernel_variant() == (0, "tuned")
    r2.set_models(1, [models[1]])
    y2 = r2.run(u)
    assert np.array_equal(y2, y)
    # after the batch has run: instance 0's state survives the move, instance 2 restarts as its new model
    r3 = ModelRunner(models[0], 3, lib=emu_lib, models=[models[0]] * 3)
    ya = r3.run(u[:, :, :25])
    r3.set_models(2, [models[2]])
    assert r3.batch_kernel_variant() == (0, "tuned")
    yb = r3.run(u[:, :, 25:])
    assert_close(np.concatenate([ya[:1], yb[:1]], axis=2), ref[0][0], rtol=RTOL_SAME)
    assert_close(yb[2:3], oracle_run(models[2], u[2:3, :, 25:])[0], rtol=RTOL_SAME)
    # the plain shape from the start (ACME_CONDENSE=0) is where the batch ends up
    monkeypatch.setenv("ACME_CONDENSE", "0")
    r0 = ModelRunner(models[0], 3, lib=emu_lib, models=models)
    assert r0.kernel_variant()[0] == 0
    assert np.array_equal(r0.run(u), y)


def test_emulated_mid_size_private_models_and_the_dense_fallback(emu_lib):
    """The mid-size kernel on a matrix in LDS with one model PER INSTANCE (34 unknowns): every instance's image carries its own
    sparse forms of the matrices with the batch's entries per row (csrc/acme_pack.h pack_generic `like`) -- instances whose
    matrices differ in their VALUES follow the oracle run of their own model; and an instance whose model has a FULLER row than
    the batch's sparse forms hold (an entry of fq where the batch model has none) switches the whole batch back to the dense
    matrices (GenHeader::ell = 0): still the oracle's outputs and iteration totals, for every instance."""
    import copy
    from helpers import HS, RTOL_SAME, mid_size_models
    name, m, u = [x for x in mid_size_models(more=True) if x[0] == "34 unknowns"][0]
    m.solver = HS
    u = u[:3]
    variants = [m]
    m1 = copy.deepcopy(m)                      # other values, the same pattern
    m1.subs[0].q0 = m1.subs[0].q0 * 1.01
    m1.c = m1.c * 0.99
    m1.subs[0].fq = m1.subs[0].fq * 1.02
    variants.append(m1)
    m2 = copy.deepcopy(m)                      # ... and one more entry in a row of fq, of dq and of c
    s2 = m2.subs[0]
    r, c = [(r, c) for r in range(s2.nq) for c in range(s2.nn) if s2.fq[r, c] == 0.0 and np.count_nonzero(s2.fq[r]) == np.count_nonzero(s2.fq, axis=1).max()][0]
    s2.fq[r, c] = 1e-3
    variants.append(m2)
    for models, what in ((variants[:2] + [m], "same pattern"), (variants, "a fuller row: dense fallback")):
        rr = emu_runner(emu_lib, m, 3, models=models)
        assert rr.kernel_family() == "coop", what
        y = np.concatenate([rr.run(u[:, :, :50]), rr.run(u[:, :, 50:])], axis=2)
        got = rr.report_arrays()["iters_total"].tolist()
        for k, mk in enumerate(models):
            yref, its = oracle_run(mk, u[k:k + 1])
            assert_close(y[k:k + 1], yref, rtol=RTOL_SAME)
            assert got[k] == its.tolist()[0], (what, k)
    assert np.abs(y[0] - y[1]).max() > 1e-9


def test_emulated_mid_size_non_finite_input(emu_lib, monkeypatch):
    """A non-finite input sample in one instance of a 34-unknown batch (src/ACME.jl:688-694: the reference throws; here that
    instance stops and reports the sample, its neighbours are unaffected): the mid-size kernel's instantiations -- one
    instance per wave and 16 lanes per instance on a matrix in LDS, the literal one -- agree with the lane-per-instance
    generic kernel in the report, the iteration totals (the failing solve's included) and every finite output."""
    import warnings
    from helpers import HS, mid_size_models
    name, m, u = [x for x in mid_size_models(more=True) if x[0] == "34 unknowns"][0]
    m.solver = HS
    u = u[:3, :, :80].copy()
    u[1, 0, 40] = np.inf
    got = {}
    for tag, env in (("lane per instance", {"ACME_COOP": "0"}), ("one instance per wave", {}), ("16 lanes", {"ACME_COOP_WAVE64": "0"}),
                     ("literal", {"ACME_COOP_LITERAL": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = emu_runner(emu_lib, m, 3)
        assert r.kernel_family() == ("generic" if tag == "lane per instance" else "coop")
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            y = r.run(u, check=False)
        ra = r.report_arrays()
        assert ra["first_nonfinite"].tolist() == [-1, 40, -1] and ra["n_warn"].tolist() == [0, 0, 0], tag
        assert np.isnan(y[1, :, 40:]).all() and np.isfinite(y[1, :, :40]).all() and np.isfinite(y[[0, 2]]).all(), tag
        got[tag] = (np.nan_to_num(y), ra["iters_total"].tolist())
        for k in env:
            monkeypatch.delenv(k)
    for tag, (y, its) in got.items():
        assert its == got["lane per instance"][1], tag
        np.testing.assert_allclose(y, got["lane per instance"][0], rtol=1e-9, atol=1e-12)
