"""The superover and birdie MODEL BLOCKS pinned by something outside this repository's front end.

tests/golden/mna_reference.npz holds first-principles integrations of the two schematics (full nodal
analysis from the component values, implicit trapezoidal rule, Newton to 1e-14 in 80-bit arithmetic;
tests/golden/make_mna_reference.py -- none of derive.py / ratmat.py / hostsolve.py / circuit.py).  The
oracle running the derived state-space models (tests/golden/superover_var.json, birdie_var_176k.json)
must reproduce them: that is what the reference leaves as `# TODO: further validate y`
(test/runtests.jl:727,747).  Tolerance: 1e-8 * max(1, |y|) with both sides at set_resabstol!(1e-13)
(measured: <= 8.8e-10; at the default 1e-10 the solver's stopping rule alone moves y by up to 9e-7, the
sensitivity DESIGN.md 3 documents).  The HIP path against the same fixture: tests/test_gpu_mna.py."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load

MNA_BOUND = 1e-8


def mna_cases():
    """(fixture name, label, u [nu, T], y_ref [T]) for every committed first-principles run."""
    d = np.load(os.path.join(GOLDEN, "mna_reference.npz"))
    cases = []
    for pots, y in zip(d["superover_pots"], d["superover_y"]):
        u = np.zeros((4, len(y)))
        u[0] = d["superover_u"]
        u[1:] = pots[:, None]
        cases.append(("superover_var", "superover pots (%g, %g, %g)" % tuple(pots), u, y))
    for (vol, amp), y in zip(d["birdie_vol_amp"], d["birdie_y"]):
        u = np.zeros((2, len(y)))
        u[0] = amp * d["birdie_u"]
        u[1] = vol
        cases.append(("birdie_var_176k", "birdie vol %g amplitude %g" % (vol, amp), u, y))
    return cases


def mna_err(y, yref):
    return float(np.abs(y - yref).max() / max(1.0, np.abs(yref).max()))


def test_fixture_shape():
    cases = mna_cases()
    assert sum(c[0] == "superover_var" for c in cases) >= 3 and sum(c[0] == "birdie_var_176k" for c in cases) >= 2
    for name, label, u, y in cases:
        assert len(y) >= 4410 and np.isfinite(y).all() and np.abs(y).max() > 0.1, label


@pytest.mark.parametrize("case", range(6))
def test_oracle_reproduces_first_principles(case):
    from oracle.refpy import RefRunner
    name, label, u, yref = mna_cases()[case]
    r = RefRunner(load(name))
    r.set_resabstol(1e-13)
    y = r.run(u)[0]
    err = mna_err(y, yref)
    print(f"{label}: oracle vs first-principles MNA {err:.2e} (bound {MNA_BOUND:.0e}), {r.report.iters_total / len(yref):.2f} iterations per sample")
    assert r.report.n_warn == 0
    assert err < MNA_BOUND, (label, err)


def test_default_tolerance_is_solver_limited():
    """At the reference's default residual tolerance (1e-10, src/solvers.jl:175) the same comparison is bounded by
    tolerance x circuit sensitivity, not by the model: 1000 x the tolerance, ~1000 x the error."""
    from oracle.refpy import RefRunner
    name, label, u, yref = mna_cases()[0]
    r = RefRunner(load(name))
    y = r.run(u)[0]
    err = mna_err(y, yref)
    print(f"{label}: oracle at the default tolerance vs MNA {err:.2e}")
    assert err < 8e-6        # the headline parity bound of tests/test_gpu_headline.py
    assert err > MNA_BOUND   # (if this ever fails the note above is out of date)


def test_emulated_kernels_reproduce_first_principles(emu_lib):
    """The kernel source (CPU wave emulator) on the same fixture: the condensed 16-lane kernel for superover, the 16-lane
    kernel for the birdie, both at 1e-13."""
    from acme_jl_amd.runner import ModelRunner
    for name in ("superover_var", "birdie_var_176k"):
        cs = [c for c in mna_cases() if c[0] == name]
        T = 400 if name == "superover_var" else 1000       # the emulator is slow; the GPU test runs the full length
        u = np.stack([c[2][:, :T] for c in cs])
        r = ModelRunner(load(name), u.shape[0], lib=emu_lib)
        r.set_resabstol(1e-13)
        y = r.run(u)
        for i, c in enumerate(cs):
            err = mna_err(y[i, 0], c[3][:T])
            print(f"{c[1]}: emulated kernel vs MNA {err:.2e} over {T} samples")
            assert err < MNA_BOUND, (c[1], err)
