"""Small test circuits transcribed from the reference's test/runtests.jl (cited per builder)."""
from fractions import Fraction

import numpy as np

from acme_jl_amd.circuit import (Circuit, bjt, capacitor, currentprobe, currentsource, diode,
                                 inductor, mosfet, opamp, opamp_macak, resistor, transformer,
                                 transformer_ja, voltageprobe, voltagesource)
from acme_jl_amd.examples import build


def bjt_test_circuit(typ, **kw):
    """test/runtests.jl:489-498 / :519-528 -- outputs (ve, vc, ie, ic), inputs (ib, vce)."""
    return build([
        ("t", bjt(typ, **kw), {}),
        ("isrc", currentsource(), {"+": ("t", "base")}),
        ("vsrc", voltagesource(), {"-": ("isrc", "-")}),
        ("veprobe", voltageprobe(), {"+": ("t", "base"), "-": ("isrc", "-")}),
        ("vcprobe", voltageprobe(), {"+": ("t", "base"), "-": ("vsrc", "+")}),
        ("ieprobe", currentprobe(), {"+": ("t", "emitter"), "-": ("isrc", "-")}),
        ("icprobe", currentprobe(), {"+": ("t", "collector"), "-": ("vsrc", "+")}),
    ])


def bjt_test_input(typ, N=100):
    ib = 1e-3 if typ == "npn" else -1e-3
    return np.vstack([np.linspace(0, ib, N),
                      np.concatenate([np.linspace(1, -1, N // 2), np.linspace(-1, 1, N // 2)])])


def mosfet_test_circuit(typ, **kw):
    """test/runtests.jl:592-597 -- inputs (vgs, vds), output id."""
    return build([
        ("vgs", voltagesource(), {"-": "gnd"}),
        ("vds", voltagesource(), {"-": "gnd"}),
        ("J", mosfet(typ, **kw), {"gate": ("vgs", "+"), "drain": ("vds", "+")}),
        ("out", currentprobe(), {"+": ("J", "source"), "-": "gnd"}),
    ])


def macak_test_circuit():
    """test/runtests.jl:652-656"""
    return build([
        ("input", voltagesource(), {"-": "gnd"}),
        ("op", opamp_macak(100, -3, 4), {"in+": ("input", "+"), "in-": [("op", "out-"), "gnd"]}),
        ("output", voltageprobe(), {"+": ("op", "out+"), "-": "gnd"}),
    ])


def ja_inductor_circuit():
    """test/runtests.jl:433-440"""
    return build([
        ("Jin", voltagesource(), {}),
        ("Jout1", currentprobe(), {"+": ("Jin", "+")}),
        ("Jout2", currentprobe(), {"+": ("Jin", "+")}),
        ("L_JA", inductor(ja=True), {1: ("Jout1", "-"), 2: ("Jin", "-")}),
        ("L_lin", inductor(174e-3), {1: ("Jout2", "-"), 2: ("Jin", "-")}),
    ])


def resistor_diode_circuit():
    """test/runtests.jl:70-86: expected y = 25e-3*log(i/is+1)."""
    i, r, is_ = 1e-3, 10e3, 1e-12
    v_d = 25e-3 * np.log(i / is_ + 1)
    c = build([
        ("vsrc", voltagesource(i * r + v_d), {"+": "supply voltage", "-": "gnd"}),
        ("r1", resistor(r), {}),
        ("d", diode(is_=is_), {"-": "gnd", "+": ("r1", 2)}),
        ("vprobe", voltageprobe(), {"-": "gnd", "+": ("r1", 2)}),
    ])
    c.connect(("r1", 1), "supply voltage")
    return c, v_d


def constant_source_clipper(v=0.8):
    """A diode clipper fed by a CONSTANT voltage source: a nonlinear model with a state but no
    inputs (nn = 2, np = 1, nx = 1, nu = 0), run in the reference as run!(model, zeros(0, T))."""
    return build([
        ("j_in", voltagesource(v), {"-": "gnd"}),
        ("r1", resistor(1e3), {1: ("j_in", "+")}),
        ("c1", capacitor(47e-9), {1: ("r1", 2), 2: "gnd"}),
        ("d1", diode(is_=1e-15), {"-": "gnd", "+": ("r1", 2)}),
        ("d2", diode(is_=1.8e-15), {"-": ("r1", 2), "+": "gnd"}),
        ("j_out", voltageprobe(), {"-": "gnd", "+": ("r1", 2)}),
    ])


def no_solution_circuit():
    """test/runtests.jl:170-176: diode driven by a current source."""
    c = Circuit()
    c.add("d", diode())
    c.add("src", currentsource())
    c.add("probe", voltageprobe())
    c.connect(("src", "+"), ("d", "+"), ("probe", "+"))
    c.connect(("src", "-"), ("d", "-"), ("probe", "-"))
    return c


def series_diodes_circuit():
    """test/runtests.jl:268-280"""
    c = build([
        ("src1", voltagesource(), {}),
        ("probe1", currentprobe(), {}),
        ("d1", diode(), {"+": ("src1", "+")}),
        ("d2", diode(), {"+": ("d1", "-"), "-": ("probe1", "+")}),
    ])
    c.connect(("probe1", "-"), ("src1", "-"))
    c.add("src2", voltagesource())
    c.add("probe2", currentprobe())
    c.add("d3", diode())
    c.connect(("src2", "+"), ("d3", "+"))
    c.connect(("d3", "-"), ("probe2", "+"))
    c.connect(("probe2", "-"), ("src2", "-"))
    return c


def opamp_shelving_circuit(Amax, GBP):
    """test/runtests.jl:629-636"""
    return build([
        ("input", voltagesource(), {"-": "gnd"}),
        ("op", opamp(maxgain=Amax, gain_bw_prod=GBP), {"in+": ("input", "+"), "out-": "gnd"}),
        ("r1", resistor(109e3), {1: ("op", "out+"), 2: ("op", "in-")}),
        ("r2", resistor(1e3), {1: ("op", "in-")}),
        ("c", capacitor(22e-9), {1: ("r2", 2), 2: "gnd"}),
        ("output", voltageprobe(), {"+": ("op", "out+"), "-": "gnd"}),
    ])


def two_stage_clipper(bias=0.0):
    """Two diode-clipper stages separated by an ideal op-amp buffer: the nonlinearity decomposes into
    two sub-problems (one per stage), the second fed by the first through the buffer -- a small model
    with nsub = 2 and a regular (I - a), for steadystate / linearize on decomposed models.  ``bias``:
    a DC source in series with the input, so that the steady state is not the origin."""
    from acme_jl_amd.circuit import capacitor, diode, opamp, resistor, voltageprobe, voltagesource
    from acme_jl_amd.examples import build
    return build([
        ("j_in", voltagesource(), {"-": "gnd"}),
        ("j_b", voltagesource(bias), {"-": ("j_in", "+")}),
        ("r1", resistor(1e3), {1: ("j_b", "+")}),
        ("c1", capacitor(47e-9), {1: ("r1", 2), 2: "gnd"}),
        ("d1", diode(is_=1e-15), {"-": "gnd", "+": ("r1", 2)}),
        ("d2", diode(is_=1.8e-15), {"-": ("r1", 2), "+": "gnd"}),
        ("buf", opamp(), {"in+": ("r1", 2), "in-": "bo", "out+": "bo", "out-": "gnd"}),
        ("r2", resistor(2.2e3), {1: "bo"}),
        ("c2", capacitor(22e-9), {1: ("r2", 2), 2: "gnd"}),
        ("d3", diode(is_=4e-9, eta=2), {"-": "gnd", "+": ("r2", 2)}),
        ("d4", diode(is_=3e-9, eta=2), {"-": ("r2", 2), "+": "gnd"}),
        ("j_out", voltageprobe(), {"-": "gnd", "+": ("r2", 2)}),
    ])


def clipper_chain(stages, tail=False, symmetric=False):
    """acme_jl_amd.examples.clipper_chain (bench.py's mid-size workload is the same circuit)"""
    from acme_jl_amd.examples import clipper_chain as make
    return make(stages, tail, symmetric)


def buffered_clipper_chain(stages, bias=0.0):
    """`stages` diode-clipper stages separated by ideal op-amp buffers (two_stage_clipper, continued): the
    nonlinearity decomposes into one sub-problem per stage, each fed by the one before through its buffer -- models
    with any number of nonlinear sub-problems."""
    from acme_jl_amd.circuit import capacitor, diode, opamp, resistor, voltageprobe, voltagesource
    from acme_jl_amd.examples import build
    spec = [("j_in", voltagesource(), {"-": "gnd"}),
            ("j_b", voltagesource(bias), {"-": ("j_in", "+")})]
    prev = ("j_b", "+")
    for k in range(stages):
        r, c, d1, d2, buf = f"r{k}", f"c{k}", f"da{k}", f"db{k}", f"buf{k}"
        spec += [(r, resistor(1e3 * (1 + 0.4 * k)), {1: prev}),
                 (c, capacitor(47e-9 / (1 + 0.3 * k)), {1: (r, 2), 2: "gnd"}),
                 (d1, diode(is_=1e-15 * (1 + k), eta=1 + 0.1 * k), {"-": "gnd", "+": (r, 2)}),
                 (d2, diode(is_=1.8e-15 * (1 + k), eta=1 + 0.1 * k), {"-": (r, 2), "+": "gnd"})]
        if k + 1 < stages:
            spec.append((buf, opamp(), {"in+": (r, 2), "in-": f"bo{k}", "out+": f"bo{k}", "out-": "gnd"}))
            prev = f"bo{k}"
        else:
            prev = (r, 2)
    spec.append(("j_out", voltageprobe(), {"-": "gnd", "+": prev}))
    return build(spec)
