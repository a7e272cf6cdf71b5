"""Host front end (exact-rational derivation) against the reference's structural pins and
analytic known answers (run through the CPU oracle)."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import circuits
from acme_jl_amd import examples
from acme_jl_amd.circuit import (Circuit, currentprobe, currentsource, resistor, topomat,
                                 voltageprobe, voltagesource)
from acme_jl_amd.model import DiscreteModel
from helpers import GOLDEN, sine
from oracle.refpy import RefRunner

FS = Fraction(1, 44100)


def test_topomat():
    """test/runtests.jl:12-21"""
    tv, ti = topomat([[1, -1, 1], [-1, 1, -1]])
    prod = np.array(tv) @ np.array(ti).T
    assert not prod.any()
    assert topomat([[0], [0]]) == ([[1]], [])
    tv, ti = topomat([[1], [-1]])
    assert tv == [] and ti == [[1]]


def test_np_pins_of_example_models():
    """np(model, k) pins: test/runtests.jl:699,724,734,744,777."""
    assert DiscreteModel(examples.diodeclipper(), FS).np(1) == 1
    assert DiscreteModel(examples.birdie(vol=0.8), FS).np(1) == 2
    assert DiscreteModel(examples.birdie(), FS).np(1) == 3
    assert DiscreteModel(examples.superover(1.0, 1.0, 1.0), FS).np(1) == 5
    assert DiscreteModel(examples.superover(), FS).np(1) == 11


def test_np_pins_of_simplified_superover():
    """test/runtests.jl:751-759,768,782-791."""
    c = examples.superover(1.0, 1.0, 1.0)
    c.add("vbsrc", voltagesource(4.5))
    c.connect(("vbsrc", "+"), "vb")
    c.connect(("vbsrc", "-"), "gnd")
    m = DiscreteModel(c, Fraction(1 / 44100))
    assert [m.np(k) for k in (1, 2, 3)] == [2, 1, 2]
    assert DiscreteModel(c, Fraction(1 / 44100), decompose_nonlinearity=False).np(1) == 5
    c = examples.superover()
    c.add("vbsrc", voltagesource(4.5))
    c.connect(("vbsrc", "+"), "vb")
    c.connect(("vbsrc", "-"), "gnd")
    m = DiscreteModel(c, Fraction(1 / 44100))
    assert [m.np(k) for k in (1, 2, 3, 4)] == [2, 2, 2, 4]


def test_nonlinearity_decomposition_order():
    """test/runtests.jl:267-292: the single diode is extracted first."""
    c = circuits.series_diodes_circuit()
    m = DiscreteModel(c, Fraction(1), decompose_nonlinearity=False)
    assert m.nn(1) == 3
    y = RefRunner(m).run(np.array([[2.0], [1.0]]))
    want = 1e-12 * (np.exp(1 / 25e-3) - 1)
    np.testing.assert_allclose(y[:, 0], [want, want], rtol=1.5e-8)
    m = DiscreteModel(c, Fraction(1))
    assert m.nn(1) == 1 and m.nn(2) == 2
    y = RefRunner(m).run(np.array([[2.0], [1.0]]))
    np.testing.assert_allclose(y[:, 0], [want, want], rtol=1.5e-8)


def test_resistor_diode_dc_point():
    """test/runtests.jl:70-86"""
    c, v_d = circuits.resistor_diode_circuit()
    y = RefRunner(DiscreteModel(c, Fraction(1))).run(np.zeros((0, 1)))
    np.testing.assert_allclose(y[0, 0], v_d, rtol=1.5e-8)


def test_reconnection_and_deletion():
    """test/runtests.jl:102-150"""
    c = Circuit()
    c.add("r1", resistor(10))
    c.add("r2", resistor(100))
    c.connect(("r2", 1), ("r1", 1))
    c.connect(("r2", 2), ("r1", 2))
    c.add("src", voltagesource(1))
    c.connect(("src", "-"), ("r1", 2))
    c.add("probe", currentprobe())
    c.connect(("probe", "+"), ("src", "+"))
    c.connect(("probe", "-"), ("r1", 1))

    def cur():
        return RefRunner(DiscreteModel(c, Fraction(1))).run(np.zeros((0, 1)))[0, 0]
    np.testing.assert_allclose(cur(), 1 / 10 + 1 / 100)
    c.disconnect(("r2", 1))
    np.testing.assert_allclose(cur(), 1 / 10)
    c.disconnect(("r1", 2))
    np.testing.assert_allclose(cur(), 0, atol=1e-15)
    c.connect(("r1", 2), ("r2", 1))
    np.testing.assert_allclose(cur(), 1 / (10 + 100))


def test_sources_and_probes_with_internal_resistance():
    """test/runtests.jl:386-429"""
    def run(spec, u):
        return RefRunner(DiscreteModel(examples.build(spec), Fraction(1))).run(u)[0, 0]
    z = np.zeros((0, 1))
    want = 100000 * 100e-3
    np.testing.assert_allclose(run([("src", currentsource(100e-3, gp=Fraction(1, 100000)), {}),
                                    ("probe", voltageprobe(), {"+": ("src", "+"), "-": ("src", "-")})], z), want)
    np.testing.assert_allclose(run([("src", currentsource(gp=Fraction(1, 100000)), {}),
                                    ("probe", voltageprobe(), {"+": ("src", "+"), "-": ("src", "-")})],
                                   np.array([[100e-3]])), want)
    np.testing.assert_allclose(run([("src", voltagesource(10, rs=100000), {}),
                                    ("probe", currentprobe(), {"+": ("src", "+"), "-": ("src", "-")})], z),
                               10 / 100000)


def test_opamp_transfer_function():
    """test/runtests.jl:626-650 (FFT of the impulse response vs the bilinear-warped H)."""
    for Amax in (10, np.inf):
        for GBP in (50e3, np.inf):
            m = DiscreteModel(circuits.opamp_shelving_circuit(Amax, GBP), Fraction(1 / 44100))
            u = np.zeros((1, 4096))
            u[0, 0] = 1
            y = RefRunner(m).run(u)[0]
            Y = np.fft.rfft(y)
            k = np.arange(len(Y))
            w = 2 * 44100 * np.tan(np.pi * k / len(y))
            s = 1j * w
            Ginv = np.sqrt(1 - 1 / Amax ** 2) * s / (2 * np.pi * GBP) + 1 / Amax
            H = (1e3 * 22e-9 * s + 1) / ((109e3 + 1e3) * 22e-9 * s + 1)
            Yref = 1 / (Ginv + H)
            # Julia's `Y ≈ Yref` on vectors is norm based
            assert np.linalg.norm(Y - Yref) <= 1.5e-8 * max(np.linalg.norm(Y), np.linalg.norm(Yref))


@pytest.mark.parametrize("typ", ["npn", "pnp"])
def test_bjt_ebers_moll(typ):
    """test/runtests.jl:489-510, atol 1e-10"""
    isc, ise, etac, etae, bf, br = 1e-6, 2e-6, 1.1, 1.0, 100, 10
    m = DiscreteModel(circuits.bjt_test_circuit(typ, isc=isc, ise=ise, etac=etac, etae=etae, bf=bf, br=br),
                      Fraction(1))
    out = RefRunner(m).run(circuits.bjt_test_input(typ))
    if typ == "pnp":
        out = -out
    ve, vc, ie, ic = out
    np.testing.assert_allclose(ie, ise * (np.exp(ve / (etae * 25e-3)) - 1)
                               - br / (1 + br) * isc * (np.exp(vc / (etac * 25e-3)) - 1), atol=1e-10, rtol=0)
    np.testing.assert_allclose(ic, -bf / (1 + bf) * ise * (np.exp(ve / (etae * 25e-3)) - 1)
                               + isc * (np.exp(vc / (etac * 25e-3)) - 1), atol=1e-10, rtol=0)


@pytest.mark.parametrize("typ", ["npn", "pnp"])
def test_bjt_gummel_poon(typ):
    """test/runtests.jl:514-546 (a 16-combination subset of the 2^8 parameter grid)."""
    isc, ise, etac, etae, bf, br = 1e-6, 2e-6, 1.1, 1.0, 100, 10
    for bits in range(0, 256, 17):
        ile = 50e-9 if bits & 1 else 0
        ilc = 100e-9 if bits & 2 else 0
        etacl = 1.2 if bits & 4 else etac
        etael = 1.1 if bits & 8 else etae
        vaf = 10 if bits & 16 else np.inf
        var = 50 if bits & 32 else np.inf
        ikf = 50e-3 if bits & 64 else np.inf
        ikr = 500e-3 if bits & 128 else np.inf
        m = DiscreteModel(circuits.bjt_test_circuit(
            typ, isc=isc, ise=ise, etac=etac, etae=etae, bf=bf, br=br, ile=ile, ilc=ilc,
            etacl=etacl, etael=etael, vaf=vaf, var=var, ikf=ikf, ikr=ikr), Fraction(1))
        out = RefRunner(m).run(circuits.bjt_test_input(typ))
        if typ == "pnp":
            out = -out
        ve, vc, ie, ic = out
        i_f = bf / (1 + bf) * ise * (np.exp(ve / (etae * 25e-3)) - 1)
        i_r = br / (1 + br) * isc * (np.exp(vc / (etac * 25e-3)) - 1)
        icc = (2 * (1 - ve / var - vc / vaf)) / (1 + np.sqrt(1 + 4 * (i_f / ikf + i_r / ikr))) * (i_f - i_r)
        ibe = 1 / bf * i_f + ile * (np.exp(ve / (etael * 25e-3)) - 1)
        ibc = 1 / br * i_r + ilc * (np.exp(vc / (etacl * 25e-3)) - 1)
        np.testing.assert_allclose(ie, icc + ibe, atol=1e-10, rtol=0)
        np.testing.assert_allclose(ic, -icc + ibc, atol=1e-10, rtol=0)


def test_mosfet_regions_exact():
    """test/runtests.jl:590-601 (exact ==)"""
    for typ, pol in (("n", 1), ("p", -1)):
        m = DiscreteModel(circuits.mosfet_test_circuit(typ, vt=1, alpha=1e-4), Fraction(1))
        y = RefRunner(m).run(pol * np.array([[0, 1, 2, 2, 2], [5, 5, 0.5, 1, 1.5]], dtype=float))
        want = pol * np.array([0, 0, 1e-4 * (1 - 0.5 / 2) * 0.5, 1e-4 * (1 - 1 / 2) * 1, 1e-4 / 2 * 1 ** 2])
        assert np.array_equal(y[0], want)


def test_mosfet_polynomial_parameters():
    """test/runtests.jl:602-624"""
    for typ, pol in (("n", 1), ("p", -1)):
        for alpha in (1e-4, (0.0205, -0.0017)):
            for vt in (1, (1.2078, 0.3238), (-1.2454, -0.199, -0.0483)):
                m = DiscreteModel(circuits.mosfet_test_circuit(typ, vt=vt, alpha=alpha, lam=0.05), Fraction(1))
                r = RefRunner(m)
                at = alpha if isinstance(alpha, tuple) else (alpha,)
                vtt = vt if isinstance(vt, tuple) else (vt,)
                for vgs in np.linspace(0, 5, 10):
                    for vds in np.linspace(0, 5, 10):
                        y = r.run(pol * np.array([[vgs], [vds]]))[0, 0]
                        a_ = sum(c * (pol * vgs) ** k for k, c in enumerate(at))
                        vt_ = sum(c * (pol * vgs) ** k for k, c in enumerate(vtt))
                        if vgs <= vt_:
                            assert y == 0
                        elif vds <= vgs - vt_:
                            np.testing.assert_allclose(y, pol * a_ * (vgs - vt_ - vds / 2) * vds * (1 + 0.05 * vds), rtol=1.5e-8)
                        else:
                            np.testing.assert_allclose(y, pol * a_ / 2 * (vgs - vt_) ** 2 * (1 + 0.05 * vds), rtol=1.5e-8)


def test_tanh_opamp():
    """test/runtests.jl:652-662"""
    m = DiscreteModel(circuits.macak_test_circuit(), Fraction(1 / 44100))
    u = np.linspace(-1, 1, 1000)
    y = RefRunner(m).run(u[None, :])[0]
    np.testing.assert_allclose(y, 0.5 * (4 + -3) + 0.5 * (4 - -3) * np.tanh(100 / (0.5 * (4 - -3)) * u), rtol=1.5e-8)


def test_ja_inductor_state_carry_over():
    """test/runtests.jl:431-457: five consecutive run! calls on one model."""
    m = DiscreteModel(circuits.ja_inductor_circuit(), FS)
    r = RefRunner(m)
    y = r.run(np.full((1, 750), 0.1))
    # isapprox on vectors is norm based in Julia
    assert np.linalg.norm(y[0, :9] - y[1, :9]) <= 1e-2 * max(np.linalg.norm(y[0, :9]), np.linalg.norm(y[1, :9]))
    assert (y[0] < y[1]).all()
    r.run(np.full((1, 500), 0.1))
    y = r.run(np.full((1, 750), 0.1))
    assert (y[0] > y[1]).all()
    y = r.run(np.full((1, 2000), -0.1))
    assert y[0, -1] < -2e-3
    y = r.run(np.zeros((1, 1000)))
    assert y[0, 0] < -2e-3
    assert np.linalg.norm(y[0] - y[0, 0]) <= 1.5e-8 * np.linalg.norm(y[0])


def test_fixtures_reproduce():
    """tests/golden/*.json are exactly what the front end derives today."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_models", os.path.join(GOLDEN, "make_models.py"))
    mm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mm)
    for name in ("diodeclipper", "superover_var", "birdie_var", "rc_ladder"):
        fresh = mm.MODELS[name]().tune_row_order().to_dict()
        stored = json.load(open(os.path.join(GOLDEN, name + ".json")))
        assert json.loads(json.dumps(fresh)) == stored, name
