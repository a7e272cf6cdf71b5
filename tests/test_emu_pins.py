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
