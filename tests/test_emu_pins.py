"""GPU-less run of the unit-level solver pins (tests/solver_pins.py) on the CPU wave emulator:
the same kernel source, the same C-ABI host code; the -m gpu twin is tests/test_gpu_pins.py."""
import solver_pins


def test_emulated_reference_lu_cases(emu_lib):
    solver_pins.check_reference_lu_cases(emu_lib)


def test_emulated_pivot_sweep(emu_lib):
    solver_pins.check_pivot_sweep(emu_lib, n=8, count=16)
    solver_pins.check_pivot_sweep(emu_lib, n=4, count=20, seed=9, per_instance=True)


def test_emulated_parabola(emu_lib):
    solver_pins.check_parabola(emu_lib)


def test_emulated_extrapolation_jacobian(emu_lib):
    solver_pins.check_extrapolation_jacobian(emu_lib)


def test_emulated_decomposed_steadystate_linearize(emu_lib):
    solver_pins.check_decomposed_analysis(emu_lib)


def test_emulated_initial_solution_on_device(emu_lib):
    """derive_batch(init_on_device=...): the construction-time solves (initial_solution,
    src/ACME.jl:453-464, and the folded constant sub-problem) of every instance in one batched
    solve through the C ABI, against the host (numpy) restatement of the same homotopy."""
    import numpy as np
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.montecarlo import derive_batch
    make = lambda value: examples.superover(1.0, 1.0, 1.0, value=value)     # noqa: E731
    nominal = {}
    make(lambda name, v: nominal.setdefault(name, v))
    rng = np.random.Generator(np.random.PCG64(7))
    vals = {k: v * (1 + 0.05 * rng.uniform(-1, 1, 9)) for k, v in nominal.items()}
    host = derive_batch(make, Fraction(1, 44100), vals)
    info = dict(lib=emu_lib)
    dev = derive_batch(make, Fraction(1, 44100), vals, init_on_device=info)
    assert info["solved"] == 2 * 10 and not info.get("fallbacks")     # 9 instances + the structure instance, 2 sub-problems
    for key in ("init_zs", "q0s"):
        for a, b in zip(host.d[key], dev.d[key]):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(host.d["x0"], dev.d["x0"], rtol=1e-9, atol=1e-15)
