"""pytest -m gpu: the HIP path against the first-principles MNA fixture (tests/test_mna_reference.py has the
why): superover with the pots as inputs (the headline model, condensed kernel) and the birdie at 176.4 kHz,
device-resident and through host buffers, at set_resabstol!(1e-13); and the headline solver stack
(HomotopySolver{CachingSolver{SimpleSolver}}) at its default tolerance within the solver-limited bound."""
import numpy as np
import pytest

from helpers import load
from test_mna_reference import MNA_BOUND, mna_cases, mna_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["superover_var", "birdie_var_176k"])
def test_hip_path_reproduces_first_principles(hip_lib, name):
    from acme_jl_amd.runner import ModelRunner
    cs = [c for c in mna_cases() if c[0] == name]
    u = np.stack([c[2] for c in cs])
    r = ModelRunner(load(name), u.shape[0], lib=hip_lib)
    r.set_resabstol(1e-13)
    y = r.run(u)
    rep = r.report_arrays()
    assert int(rep["n_warn"].sum()) == 0
    for i, c in enumerate(cs):
        err = mna_err(y[i, 0], c[3])
        print(f"{c[1]}: HIP ({r.kernel_shape()}) vs first-principles MNA {err:.2e} (bound {MNA_BOUND:.0e}), "
              f"{rep['iters_total'][i] / u.shape[2]:.2f} iterations per sample")
        assert err < MNA_BOUND, (c[1], err)


def test_headline_stack_against_first_principles(hip_lib):
    """The bench's solver stack at the reference's default tolerance: within tolerance x circuit sensitivity of the
    first-principles run (the bound of test_gpu_headline.py), and at 1e-13 within the model bound."""
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    cs = [c for c in mna_cases() if c[0] == "superover_var"]
    u = np.stack([c[2] for c in cs])
    for tol, bound in ((None, 8e-6), (1e-13, MNA_BOUND)):
        r = ModelRunner(load("superover_var", CachingHomotopySolver), u.shape[0], lib=hip_lib)
        if tol is not None:
            r.set_resabstol(tol)
        y = r.run(u)
        errs = [mna_err(y[i, 0], c[3]) for i, c in enumerate(cs)]
        print(f"caching stack, tol {tol or 1e-10:g}: vs first-principles MNA {max(errs):.2e} (bound {bound:.0e})")
        assert max(errs) < bound
