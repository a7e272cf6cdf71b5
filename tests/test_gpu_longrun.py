"""pytest -m gpu: the regimes bench.py actually times for BASELINE configs 4 and 5 -- whole seconds of audio, not their
first milliseconds -- against the oracle, as tests/test_gpu_headline.py does for config 3; and the reference's full
Gummel-Poon known-answer grid (test/runtests.jl:514-546) on the HIP path."""
from fractions import Fraction

import numpy as np
import pytest

from helpers import FS, RTOL, RTOL_SAME, load, sine
from test_gpu_headline import oracle_parallel, rel_err

pytestmark = pytest.mark.gpu


def test_config5_long_run(hip_lib):
    """BASELINE config 5 at its own length: 16 instances spread over bench.py's amplitude x vol grid, 176 400 samples
    (1 s at 176.4 kHz) of the birdie with vol as an input, HomotopySolver{SimpleSolver} on both sides -- the same Newton
    paths, hence RTOL_SAME-level agreement where the iteration totals are identical, RTOL otherwise."""
    import bench
    from acme_jl_amd.runner import ModelRunner
    N, T = 2048, 176400
    _, vol, amp = bench.grid_inputs("birdie_grid", 0, 1, N, T)
    idx = np.linspace(0, N - 1, 16).astype(int)
    u = np.zeros((16, 2, T))
    u[:, 0] = amp[idx, None] * sine(T, fs=176400)[None, :]
    u[:, 1] = vol[idx]
    m = load("birdie_var_176k")
    r = ModelRunner(m, 16, lib=hip_lib)
    y = r.run(u)
    ra = r.report_arrays()
    yo, ito, wo = oracle_parallel("birdie_var_176k", None, u)
    err = rel_err(y, yo)
    same = int((ra["iters_total"] == ito).sum())
    print(f"config 5, 16 x {T} samples: GPU vs oracle {err:.2e}; iterations GPU {ra['iters_total'].sum()} oracle {ito.sum()} "
          f"({same}/16 instances identical), {ra['iters_total'].sum() / (16 * T):.3f} per sample; warnings {ra['n_warn'].sum()} / {wo.sum()}")
    assert (ra["first_nonfinite"] < 0).all() and ra["n_warn"].sum() == wo.sum() == 0
    assert err <= (RTOL_SAME if same == 16 else RTOL)
    assert abs(int(ra["iters_total"].sum()) - int(ito.sum())) <= 1e-3 * ito.sum()


def _config4_models(n):
    from acme_jl_amd import examples
    from acme_jl_amd.montecarlo import derive_batch
    make = lambda value: examples.superover(1.0, 1.0, 1.0, value=value)     # noqa: E731
    nominal = {}
    make(lambda name, v: nominal.setdefault(name, v))
    rng = np.random.Generator(np.random.PCG64([20250905, 0]))
    vals = {k: v * (1 + 0.05 * rng.uniform(-1, 1, 8192))[:n] for k, v in nominal.items()}    # the bench's first n instances
    return make, vals, derive_batch(make, Fraction(1, FS), vals)


def _oracle_exact_job(args):
    k, vals_k, u, limit, tol, solver = args
    from acme_jl_amd import examples
    from acme_jl_amd.model import DiscreteModel
    from oracle.refpy import RefRunner
    exact = DiscreteModel(examples.superover(1.0, 1.0, 1.0, value=lambda name, v: vals_k[name]), Fraction(1, FS), solver=solver)
    r = RefRunner(exact)
    if limit is not None:
        r.set_cache_limit(limit)
    if tol is not None:
        r.set_resabstol(tol)
    y = r.run(u)
    return y, r.report.iters_total, r.report.n_warn


def test_config4_long_run(hip_lib):
    """BASELINE config 4 at its own length: 8 Monte-Carlo instances (the bench's first 8: PCG64 seed 20250905, every
    resistor / capacitor / pot track +-5 %) x 44 100 samples on the reference's default caching stack, private model
    blocks from the structure-replaying front end on the GPU, against oracle runs of the EXACTLY derived per-instance
    models with the bounded (16) and the reference's unbounded solution store -- and all three against the ROOT (the exact
    models' oracle at set_resabstol!(1e-15)): at the default tolerance the distance is solver tolerance x circuit
    sensitivity for the reference's own stores as for the GPU (tests/test_gpu_headline.py has the headline's figures); at
    1e-13 it follows the tolerance down."""
    import multiprocessing as mp
    from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver
    from acme_jl_amd.runner import ModelRunner
    n, T = 8, FS
    make, vals, batch = _config4_models(n)
    batch.solver = CachingHomotopySolver
    u = np.tile(sine(T)[None, None, :], (n, 1, 1))
    r = ModelRunner(batch.model(0), n, models=batch, lib=hip_lib)
    y = r.run(u)
    ra = r.report_arrays()
    assert (ra["n_warn"] == 0).all() and (ra["first_nonfinite"] < 0).all()
    rt = ModelRunner(batch.model(0), n, models=batch, lib=hip_lib)
    rt.set_resabstol(1e-13)
    yt = rt.run(u)
    from oracle import refpy
    refpy.lib()
    legs = [(16, None, CachingHomotopySolver), (0, None, CachingHomotopySolver), (None, 1e-15, HomotopySolver), (0, 1e-13, CachingHomotopySolver)]
    jobs = [(k, {name: float(v[k]) for name, v in vals.items()}, u[k], lim, tol, solver) for lim, tol, solver in legs for k in range(n)]
    with mp.get_context("fork").Pool(min(len(jobs), 16)) as pool:
        res = pool.map(_oracle_exact_job, jobs, chunksize=1)
    yb, yu, yroot, yut = (np.stack([x[0] for x in res[i * n:(i + 1) * n]]) for i in range(4))
    itb = np.array([x[1] for x in res[:n]])
    assert sum(x[2] for x in res) == 0
    eb, eu, ebu = rel_err(y, yb), rel_err(y, yu), rel_err(yb, yu)
    c_gpu, c_ob, c_ou = (rel_err(v, yroot) / 1e-10 for v in (y, yb, yu))
    et = rel_err(yt, yut)
    print(f"config 4, 8 x {T} samples: GPU vs exact oracle(16) {eb:.2e}, vs oracle(unbounded) {eu:.2e}, oracle(16) vs "
          f"oracle(unbounded) {ebu:.2e}; distance from the root in units of tol = 1e-10: GPU {c_gpu:.3g}, oracle(16) {c_ob:.3g}, "
          f"oracle(unbounded) {c_ou:.3g}; at tol 1e-13 GPU vs oracle(unbounded) {et:.2e}; iterations GPU {ra['iters_total'].sum()} "
          f"oracle(16) {itb.sum()} ({ra['iters_total'].sum() / (n * T):.3f} per sample)")
    assert max(c_gpu, c_ob, c_ou) <= 1.5e5          # the circuit's sensitivity, measured x 1.5
    assert c_gpu <= 1.5 * max(c_ob, c_ou)           # the GPU is no further from the truth than the reference's own stores
    assert eb <= 1.5e-5 and eu <= 1.5e-5
    assert et <= 1.5e-8
    assert abs(int(ra["iters_total"].sum()) - int(itb.sum())) <= 0.02 * itb.sum()


def _gp_params(bits):
    etac, etae = 1.1, 1.0
    return dict(ile=50e-9 if bits & 1 else 0, ilc=100e-9 if bits & 2 else 0, etacl=1.2 if bits & 4 else etac,
                etael=1.1 if bits & 8 else etae, vaf=10 if bits & 16 else np.inf, var=50 if bits & 32 else np.inf,
                ikf=50e-3 if bits & 64 else np.inf, ikr=500e-3 if bits & 128 else np.inf)


def test_gummel_poon_full_grid(hip_lib):
    """test/runtests.jl:514-546 IN FULL on the HIP path: 2^8 parameter combinations x npn / pnp x 100 samples, the GPU's
    (ve, vc, ie, ic) against the closed form at the reference's own atol = 1e-10 -- as ONE batch of 512 instances with
    per-instance model blocks AND per-instance element tables (every combination is another set of closure parameters and
    another set of the closure's compile-time branches, src/elements.jl:331-396)."""
    import circuits
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    isc, ise, etac, etae, bf, br = 1e-6, 2e-6, 1.1, 1.0, 100, 10
    models, us, meta = [], [], []
    for typ in ("npn", "pnp"):
        for bits in range(256):
            p = _gp_params(bits)
            models.append(DiscreteModel(circuits.bjt_test_circuit(typ, isc=isc, ise=ise, etac=etac, etae=etae, bf=bf, br=br, **p), Fraction(1)))
            us.append(circuits.bjt_test_input(typ))
            meta.append((typ, bits, p))
    r = ModelRunner(models[-1], len(models), lib=hip_lib, models=models)
    y = r.run(np.stack(us))
    worst = 0.0
    for (typ, bits, p), out in zip(meta, y):
        if typ == "pnp":
            out = -out
        ve, vc, ie, ic = out
        i_f = bf / (1 + bf) * ise * (np.exp(ve / (etae * 25e-3)) - 1)
        i_r = br / (1 + br) * isc * (np.exp(vc / (etac * 25e-3)) - 1)
        icc = (2 * (1 - ve / p["var"] - vc / p["vaf"])) / (1 + np.sqrt(1 + 4 * (i_f / p["ikf"] + i_r / p["ikr"]))) * (i_f - i_r)
        ibe = 1 / bf * i_f + p["ile"] * (np.exp(ve / (p["etael"] * 25e-3)) - 1)
        ibc = 1 / br * i_r + p["ilc"] * (np.exp(vc / (p["etacl"] * 25e-3)) - 1)
        e = max(float(np.abs(ie - (icc + ibe)).max()), float(np.abs(ic - (-icc + ibc)).max()))
        worst = max(worst, e)
        assert e <= 1e-10, (typ, bits, e)
    print(f"Gummel-Poon grid: 512 models x 100 samples as ONE batch on the HIP path (kernel shape {r.kernel_shape()}), worst |error| vs "
          f"the closed form {worst:.2e} A (reference atol 1e-10)")


def test_per_instance_element_parameters(hip_lib):
    """A sweep over ELEMENT parameters as one batch: 64 diode clippers, every one with its own diode
    saturation currents and emission coefficients, against oracle runs of the 64 exactly derived models (RTOL_SAME,
    identical iteration totals); the emulator's mixed Gummel-Poon batch likewise."""
    import sys
    from helpers import RTOL_SAME, oracle_run, assert_close
    from acme_jl_amd import examples
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    from helpers import HS
    n, T = 64, 2000
    t = Fraction(1, FS)
    models = [DiscreteModel(examples.diodeclipper(is1=1e-15 * 10 ** (3 * k / (n - 1)), is2=1.8e-15 * 10 ** (2 * (n - 1 - k) / (n - 1)),
                                                  eta1=1 + 0.05 * (k % 3), eta2=1 + 0.04 * (k % 4)), t, HS) for k in range(n)]
    u = np.linspace(0.2, 3.0, n)[:, None, None] * sine(T)[None, None, :]
    r = ModelRunner(models[0], n, lib=hip_lib, models=models)
    y = np.concatenate([r.run(u[:, :, :700]), r.run(u[:, :, 700:])], axis=2)
    its = r.report_arrays()["iters_total"]
    worst = 0.0
    for k, m in enumerate(models):
        yref, iref = oracle_run(m, u[k:k + 1])
        worst = max(worst, assert_close(y[k:k + 1], yref, rtol=RTOL_SAME))
        assert its[k] == iref[0], k
    print(f"64 diode clippers with their own is / eta in one batch ({r.kernel_shape()}): worst rel err vs the exact models' oracle {worst:.2e}, "
          f"identical iteration totals")
    # superover with the pots as inputs (the headline model: a CONDENSED kernel shape) and per-instance diode saturation
    # currents: the library moves the batch to the plain shape by itself -- no environment variable
    from test_emu_parity import element_parameter_sweeps, superover_models_with_their_own_diodes
    from helpers import sweep_inputs
    sm = superover_models_with_their_own_diodes(20, HS)
    us = sweep_inputs("superover_var", 20, 400, seed=2)
    rs = ModelRunner(sm[0], 20, lib=hip_lib, models=sm)
    assert rs.kernel_variant()[0] > 0 and rs.batch_kernel_variant() == (0, "tuned")
    ys = np.concatenate([rs.run(us[:, :, :150]), rs.run(us[:, :, 150:])], axis=2)
    for k, m in enumerate(sm):
        yref, iref = oracle_run(m, us[k:k + 1])
        assert_close(ys[k:k + 1], yref, rtol=RTOL_SAME)
        assert rs.report_arrays()["iters_total"][k] == iref[0], k
    print("20 superover models (pots as inputs) with their own diode saturation currents: batch moved off the condensed shape by itself, "
          f"kernel shape {rs.kernel_shape()}, oracle's outputs and iteration totals")
    name, gp, ugp = element_parameter_sweeps()[1]
    rg = ModelRunner(gp[0], len(gp), lib=hip_lib, models=gp)
    yg = rg.run(ugp)
    for k, m in enumerate(gp):
        yref, iref = oracle_run(m, ugp[k:k + 1])
        assert_close(yg[k:k + 1], yref, rtol=RTOL_SAME)
        assert rg.report_arrays()["iters_total"][k] == iref[0], k
