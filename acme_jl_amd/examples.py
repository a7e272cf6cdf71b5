"""The reference's example circuits, transcribed netlist for netlist.

  diodeclipper ... examples/diodeclipper.jl:6-15
  superover ...... examples/superover.jl:11-72
  birdie ......... examples/birdie.jl:13-31
  sallenkey ...... examples/sallenkey.jl:6-17
  rc_ladder ...... docs/src/ug.md:40-56 (20-stage RC ladder of the user guide)

``build(circ, spec)`` is a tiny stand-in for the ``@circuit`` DSL
(src/circuit.jl:317-406): each entry is ``(designator, element, {pin: target, ...})``
where a target is a named net (str) or a ``(designator, pin)`` tuple.
"""
from __future__ import annotations

from .circuit import (Circuit, bjt, capacitor, diode, opamp, potentiometer, resistor,
                      voltageprobe, voltagesource)


def build(spec, circ=None):
    circ = circ or Circuit()
    for des, elem, conns in spec:
        circ.add(des, elem)
        for pin, targets in conns.items():
            if not isinstance(targets, list):
                targets = [targets]
            circ.connect((des, pin), *targets)
    return circ


def diodeclipper(is1=1e-15, is2=1.8e-15, eta1=1, eta2=1):
    """examples/diodeclipper.jl:6-15; the diodes' parameters as arguments (sweeps over element parameters)"""
    return build([
        ("j_in", voltagesource(), {"-": "gnd"}),
        ("r1", resistor(1e3), {1: ("j_in", "+")}),
        ("c1", capacitor(47e-9), {1: ("r1", 2), 2: "gnd"}),
        ("d1", diode(is_=is1, eta=eta1), {"-": "gnd", "+": ("r1", 2)}),
        ("d2", diode(is_=is2, eta=eta2), {"-": ("r1", 2), "+": "gnd"}),
        ("j_out", voltageprobe(), {"-": "gnd", "+": ("r1", 2)}),
    ])


def _pot(r, pos):
    return potentiometer(r) if pos is None else potentiometer(r, pos)


def superover(drive=None, tone=None, level=None, sym=False, value=None):
    # value(name, nominal) -> value used for resistor / capacitor / pot-track `name`
    # (component-tolerance Monte-Carlo, BASELINE config 4); default: nominal
    V = (lambda name, nominal: nominal) if value is None else value
    circ = build([
        # power supply
        ("j3", voltagesource(9), {"+": "vcc", "-": "gnd"}),
        ("d4", diode(is_=12e-9, eta=2), {"-": "vcc", "+": "gnd"}),
        ("c11", capacitor(V("c11", 100e-6)), {1: "vcc", 2: "gnd"}),
        ("r17", resistor(V("r17", 33e3)), {1: "vcc", 2: "vb"}),
        ("r18", resistor(V("r18", 33e3)), {1: "vb", 2: "gnd"}),
        ("c12", capacitor(V("c12", 47e-6)), {1: "vb", 2: "gnd"}),
        # input stage
        ("j1", voltagesource(), {"-": "gnd"}),
        ("r1", resistor(V("r1", 2.2e6)), {1: ("j1", "+"), 2: "gnd"}),
        ("c1", capacitor(V("c1", 47e-9)), {1: ("j1", "+")}),
        ("r2", resistor(V("r2", 10e3)), {1: ("c1", 2)}),
        ("r3", resistor(V("r3", 470e3)), {1: ("r2", 2), 2: "vb"}),
        ("q1", bjt("npn", is_=80e-15, bf=500, br=10), {"base": ("r2", 2), "collector": "vcc"}),
        ("r4", resistor(V("r4", 10e3)), {1: ("q1", "emitter"), 2: "gnd"}),
        ("c2", capacitor(V("c2", 18e-9)), {1: ("q1", "emitter")}),
        ("r5", resistor(V("r5", 100e3)), {1: ("c2", 2), 2: "vb"}),
        # distortion stage
        ("ic1a", opamp(), {"in+": ("c2", 2), "out-": "gnd"}),
        ("d1", diode(is_=4e-9, eta=2), {"-": ("ic1a", "out+"), "+": ("ic1a", "in-")}),
        ("d2", diode(is_=3e-9, eta=2), {"-": ("ic1a", "in-")}),
        ("d3", diode(is_=5e-9, eta=2), {"+": ("ic1a", "out+"), "-": ("d2", "+")}),
        ("p1", _pot(V("p1", 1e6), drive), {2: [("p1", 3), ("ic1a", "out+")]}),
        ("r6", resistor(V("r6", 33e3)), {1: ("ic1a", "in-"), 2: ("p1", 1)}),
        ("c4", capacitor(V("c4", 47e-9)), {1: ("ic1a", "in-")}),
        ("r7", resistor(V("r7", 4.7e3)), {1: ("c4", 2), 2: "vb"}),
        # tone control stage
        ("r8", resistor(V("r8", 10e3)), {1: ("ic1a", "out+")}),
        ("ic1b", opamp(), {"in+": ("r8", 2), "out-": "gnd"}),
        ("c5", capacitor(V("c5", 18e-9)), {1: ("ic1b", "in+"), 2: "gnd"}),
        ("r10", resistor(V("r10", 10e3)), {1: ("ic1b", "out+"), 2: ("ic1b", "in-")}),
        ("c7", capacitor(V("c7", 10e-9)), {1: ("ic1b", "out+"), 2: ("ic1b", "in-")}),
        ("p2", _pot(V("p2", 20e3), tone), {1: ("ic1b", "in+"), 3: ("ic1b", "in-")}),
        ("c6", capacitor(V("c6", 27e-9)), {1: ("p2", 2)}),
        ("r11", resistor(V("r11", 470)), {1: ("c6", 2), 2: "gnd"}),
        # output stage
        ("c8", capacitor(V("c8", 1e-3)), {1: ("ic1b", "out+")}),
        ("r12", resistor(V("r12", 4.7e3)), {1: ("c8", 2)}),
        ("p3", _pot(V("p3", 10e3), level), {1: "vb", 3: ("r12", 2)}),
        ("r20", resistor(V("r20", 22e3)), {1: ("p3", 2)}),
        ("c9", capacitor(V("c9", 47e-9)), {1: ("r20", 2)}),
        ("r13", resistor(V("r13", 1e6)), {1: ("c9", 2), 2: "vb"}),
        ("q2", bjt("npn", is_=80e-15, bf=500, br=10), {"base": ("c9", 2), "collector": "vcc"}),
        ("r14", resistor(V("r14", 10e3)), {1: ("q2", "emitter"), 2: "gnd"}),
        ("r15", resistor(V("r15", 1e3)), {1: ("q2", "emitter")}),
        ("c10", capacitor(V("c10", 1e-6)), {1: ("r15", 2)}),
        ("r16", resistor(V("r16", 100e3)), {1: ("c10", 2), 2: "gnd"}),
        ("j2", voltageprobe(), {"+": ("c10", 2), "-": "gnd"}),
    ])
    if sym:
        circ.connect(("d3", "-"), ("d3", "+"))
    return circ


def birdie(vol=None):
    return build([
        ("j3", voltagesource(9), {"-": "gnd", "+": "vcc"}),
        ("c5", capacitor(100e-6), {1: "gnd", 2: "vcc"}),
        ("d1", diode(is_=350e-12, eta=1.6), {"-": "vcc", "+": "gnd"}),
        ("j1", voltagesource(), {"-": "gnd"}),
        ("r1", resistor(1e6), {1: ("j1", "+"), 2: "gnd"}),
        ("c1", capacitor(2.2e-9), {1: ("j1", "+")}),
        ("r2", resistor(43e3), {1: ("c1", 2), 2: "gnd"}),
        ("r3", resistor(430e3), {1: ("c1", 2), 2: "vcc"}),
        ("t1", bjt("npn", isc=154.1e-15, ise=64.53e-15, etac=1.10, etae=1.06, bf=500, br=12),
         {"base": ("c1", 2)}),
        ("r4", resistor(390), {1: ("t1", "emitter"), 2: "gnd"}),
        ("r5", resistor(10e3), {1: ("t1", "collector"), 2: "vcc"}),
        ("c3", capacitor(2.2e-9), {1: ("t1", "collector")}),
        ("p1", _pot(100e3, vol), {1: "gnd", 3: ("c3", 2)}),
        ("j2", voltageprobe(), {"-": "gnd", "+": ("p1", 2)}),
    ])


def sallenkey():
    return build([
        ("j_in", voltagesource(), {"-": "gnd"}),
        ("r1", resistor(10e3), {1: ("j_in", "+")}),
        ("r2", resistor(10e3), {1: ("r1", 2)}),
        ("c1", capacitor(10e-9), {1: ("r1", 2)}),
        ("u1", opamp(), {"in+": ("r2", 2), "in-": [("u1", "out+"), ("c1", 2)], "out-": "gnd"}),
        ("c2", capacitor(10e-9), {1: ("u1", "in+"), 2: "gnd"}),
        ("j_out", voltageprobe(), {"-": "gnd", "+": ("u1", "out+")}),
    ])


def rc_ladder(nstages=20, r=1e3, c=10e-9):
    """docs/src/ug.md:40-56: source -> (R series, C to ground) x n -> probe."""
    circ = Circuit()
    circ.add("src", voltagesource())
    circ.add("output", voltageprobe())
    circ.connect(("src", "-"), ("output", "-"), "gnd")
    pin = ("src", "+")
    for i in range(1, nstages + 1):
        rr, cc = f"r{i}", f"c{i}"
        circ.add(rr, resistor(r))
        circ.add(cc, capacitor(c))
        circ.connect((rr, 1), pin)
        circ.connect((rr, 2), (cc, 1))
        circ.connect((cc, 2), "gnd")
        pin = (rr, 2)
    circ.connect(pin, ("output", "+"))
    return circ


def clipper_chain(stages, tail=False, symmetric=False):
    """`stages` diode-clipper stages in a row (R - C||diode pair, loaded by the next stage's R) -- NOT one of the
    reference's examples: a circuit whose nonlinearity does not decompose, so that undecomposed
    (`decompose_nonlinearity=False`) it is ONE nonlinear sub-problem of 2*stages unknowns: the model for the padded kernel
    shapes (up to 16 unknowns), the cooperative mid-size kernel (17 ... 64; bench.py's `clipper_chain_20`) and the
    lane-per-instance generic kernel.  tail: one more series resistor with a single diode to ground behind the last
    stage (an odd number of unknowns); symmetric: both diodes of a stage alike (Jacobian entries of equal magnitude: the
    pivot search meets ties)."""
    spec = [("j_in", voltagesource(), {"-": "gnd"})]
    prev = ("j_in", "+")
    for k in range(stages):
        r, c, d1, d2 = f"r{k}", f"c{k}", f"da{k}", f"db{k}"
        spec += [(r, resistor(1e3 * (1 + 0.3 * k)), {1: prev}),
                 (c, capacitor(47e-9 / (1 + 0.2 * k)), {1: (r, 2), 2: "gnd"}),
                 (d1, diode(is_=1e-15 * (1 + k)), {"-": "gnd", "+": (r, 2)}),
                 (d2, diode(is_=(1.0e-15 if symmetric else 1.8e-15) * (1 + k)), {"-": (r, 2), "+": "gnd"})]
        prev = (r, 2)
    if tail:
        spec += [("rt", resistor(4.7e3), {1: prev}),
                 ("dt", diode(is_=3e-15), {"-": "gnd", "+": ("rt", 2)})]
    spec.append(("j_out", voltageprobe(), {"-": "gnd", "+": prev}))
    return build(spec)
