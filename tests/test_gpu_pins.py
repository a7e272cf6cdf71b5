"""Unit-level pins of the HIP path's LinearSolver / SimpleSolver / HomotopySolver through
acme_batch_solve (bodies in tests/solver_pins.py; reference cases test/runtests.jl:23-41,
207-219).  Run on the MI355X box with `pytest -m gpu`."""
import pytest

import solver_pins

pytestmark = pytest.mark.gpu


def test_reference_lu_cases(hip_lib):
    solver_pins.check_reference_lu_cases(hip_lib)


def test_pivot_sweep(hip_lib):
    w8 = solver_pins.check_pivot_sweep(hip_lib, n=8, count=48)
    w4 = solver_pins.check_pivot_sweep(hip_lib, n=4, count=64, seed=9, per_instance=True)
    print(f"pivot sweep: worst scaled error n=8 {w8:.2e}, n=4 {w4:.2e}")


def test_parabola_homotopy(hip_lib):
    solver_pins.check_parabola(hip_lib)


def test_extrapolation_jacobian(hip_lib):
    solver_pins.check_extrapolation_jacobian(hip_lib)


def test_decomposed_steadystate_linearize(hip_lib):
    solver_pins.check_decomposed_analysis(hip_lib)


def test_initial_solution_of_65536_models_on_gpu(hip_lib):
    """BASELINE config 4's 65 536 Monte-Carlo models take their initial solutions
    (src/ACME.jl:453-464) from ONE batched GPU solve per sub-problem (derive_batch(init_on_device));
    every instance converges, and a spread sample agrees with the host restatement of the same
    homotopy to the solver tolerance."""
    import time
    import numpy as np
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.montecarlo import derive_batch
    make = lambda value: examples.superover(1.0, 1.0, 1.0, value=value)     # noqa: E731
    nominal = {}
    make(lambda name, v: nominal.setdefault(name, v))
    N = 65536
    rng = np.random.Generator(np.random.PCG64(20250905))
    vals = {k: v * (1 + 0.05 * rng.uniform(-1, 1, N)) for k, v in nominal.items()}
    info = dict(lib=hip_lib)
    t0 = time.perf_counter()
    dev = derive_batch(make, Fraction(1, 44100), vals, init_on_device=info)
    dt = time.perf_counter() - t0
    print(f"65536 models derived in {dt:.1f} s, {info['solved']} initial solutions on the GPU")
    assert info["solved"] == 2 * (N + 1) and not info.get("fallbacks")
    pick = np.linspace(0, N - 1, 64).astype(int)
    host = derive_batch(make, Fraction(1, 44100), {k: v[pick] for k, v in vals.items()})
    for a, b in zip(host.d["init_zs"], dev.d["init_zs"]):
        np.testing.assert_allclose(a, b[pick], rtol=1e-8, atol=1e-11)
    assert np.isfinite(dev.d["init_zs"][0]).all()
