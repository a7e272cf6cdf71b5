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
