#!/usr/bin/env python
"""First-principles reference for the superover and birdie circuits: full modified nodal analysis
straight from the component values of the reference's netlists, implicit trapezoidal rule, Newton
to 1e-14 in extended precision.  Writes tests/golden/mna_reference.npz.

WHY.  The reference pins the *dimensions* of these two models and leaves their sample values as
`# TODO: further validate y` (test/runtests.jl:727,747).  The oracle and the HIP kernels run model
blocks that this repository's own front-end restatement derives (acme_jl_amd/derive.py, ratmat.py,
hostsolve.py, circuit.py) -- GPU == oracle therefore says nothing about whether those blocks describe
the circuit.  This script uses NONE of that code: no Element matrices, no gensolve, no state-space
model, no nonlinear decomposition.  It writes Kirchhoff's current law for every node of the schematic
and solves it sample by sample.  tests/test_mna_reference.py (oracle) and tests/test_gpu_mna.py (HIP)
compare against the fixture.

WHAT it restates (file:line in the ACME.jl tree) -- physics and the discretisation only:
  netlists             examples/superover.jl:11-72, examples/birdie.jl:13-31 (component values, pins)
  resistor             v = r i                                   src/elements.jl:16
  potentiometer(r,pos) r pos between pins 1-2, r (1-pos) between pins 2-3   src/elements.jl:18-19
  capacitor            c v = x, i = dx/dt                         src/elements.jl:38
  diode                i = is (exp(v / (25e-3 eta)) - 1)          src/elements.jl:236-245
  bjt (Ebers-Moll)     iE = i_cc + i_f / bf, iC = -i_cc + i_r / br (ports base->emitter,
                       base->collector), i_f = bf/(1+bf) ise (exp(vE/(25e-3 eta_e)) - 1), i_r likewise
                                                                  src/elements.jl:323-334,375-401
  ideal op-amp         v(in+) = v(in-), no input current, output current free   src/elements.jl:508-517
  discretisation       every state enters as (x[n] + x[n-1]) / 2 and (x[n] - x[n-1]) / T, x[-1] = 0
                       (src/ACME.jl:265-272, :145): the implicit trapezoidal rule, i.e. for a capacitor
                       i[n] = (2c/T) v[n] - (2/T) x[n-1],  x[n] = 2 c v[n] - x[n-1]

Arithmetic: numpy longdouble (x87 80-bit, eps 1.1e-19) with an own partially pivoted Gaussian
elimination -- the nodal matrix mixes 2C/T = 88 S (c8 = 1 mF) with 1/2.2 MOhm, condition ~1e9, and the
fixture is meant to be good to 1e-12.

usage: python tests/golden/make_mna_reference.py          (about a minute)
"""
import os
import sys

import numpy as np

LD = np.longdouble
HERE = os.path.dirname(os.path.abspath(__file__))


class Netlist:
    """Nodes by name ('gnd' is the datum), elements as plain tuples."""

    def __init__(self):
        self.nodes = {"gnd": -1}
        self.R, self.C, self.V, self.OA, self.D, self.Q = [], [], [], [], [], []

    def n(self, name):
        if name not in self.nodes:
            self.nodes[name] = len(self.nodes) - 1
        return self.nodes[name]

    def resistor(self, r, a, b):
        self.R.append((LD(r), self.n(a), self.n(b)))

    def pot(self, r, pos, p1, p2, p3):
        self.resistor(LD(r) * LD(pos), p1, p2)
        self.resistor(LD(r) * (LD(1) - LD(pos)), p2, p3)

    def capacitor(self, c, a, b):
        self.C.append((LD(c), self.n(a), self.n(b)))

    def vsource(self, plus, minus, value):        # value: float, or "u" for the input signal
        self.V.append((self.n(plus), self.n(minus), value))

    def opamp(self, inp, inm, outp, outm="gnd"):
        self.OA.append((self.n(inp), self.n(inm), self.n(outp), self.n(outm)))

    def diode(self, anode, cathode, is_, eta):
        self.D.append((self.n(anode), self.n(cathode), LD(is_), LD(1) / (LD(25e-3) * LD(eta))))

    def npn(self, b, c, e, ise, isc, eta_e, eta_c, bf, br):
        bf, br = LD(bf), LD(br)
        self.Q.append((self.n(b), self.n(c), self.n(e), bf / (1 + bf) * LD(ise), br / (1 + br) * LD(isc),
                       LD(1) / (LD(25e-3) * LD(eta_e)), LD(1) / (LD(25e-3) * LD(eta_c)), bf, br))


def superover(drive, tone, level):
    """examples/superover.jl:11-72 (R19 and LED D5 are not in the reference's model either)."""
    c = Netlist()
    # power supply
    c.vsource("vcc", "gnd", 9.0)                                  # j3
    c.diode("gnd", "vcc", 12e-9, 2)                               # d4: [-] vcc, [+] gnd
    c.capacitor(100e-6, "vcc", "gnd")                             # c11
    c.resistor(33e3, "vcc", "vb")                                 # r17
    c.resistor(33e3, "vb", "gnd")                                 # r18
    c.capacitor(47e-6, "vb", "gnd")                               # c12
    # input stage
    c.vsource("in", "gnd", "u")                                   # j1
    c.resistor(2.2e6, "in", "gnd")                                # r1
    c.capacitor(47e-9, "in", "c1b")                               # c1
    c.resistor(10e3, "c1b", "q1b")                                # r2
    c.resistor(470e3, "q1b", "vb")                                # r3
    c.npn("q1b", "vcc", "q1e", 80e-15, 80e-15, 1, 1, 500, 10)     # q1
    c.resistor(10e3, "q1e", "gnd")                                # r4
    c.capacitor(18e-9, "q1e", "a_inp")                            # c2
    c.resistor(100e3, "a_inp", "vb")                              # r5
    # distortion stage
    c.opamp("a_inp", "a_inm", "a_out")                            # ic1a
    c.diode("a_inm", "a_out", 4e-9, 2)                            # d1: [-] out+, [+] in-
    c.diode("dmid", "a_inm", 3e-9, 2)                             # d2: [-] in-, [+] d3[-]
    c.diode("a_out", "dmid", 5e-9, 2)                             # d3: [+] out+, [-] d2[+]
    c.pot(1e6, drive, "p1_1", "a_out", "a_out")                   # p1: [2] = [3] = out+
    c.resistor(33e3, "a_inm", "p1_1")                             # r6
    c.capacitor(47e-9, "a_inm", "c4b")                            # c4
    c.resistor(4.7e3, "c4b", "vb")                                # r7
    # tone control stage
    c.resistor(10e3, "a_out", "b_inp")                            # r8
    c.opamp("b_inp", "b_inm", "b_out")                            # ic1b
    c.capacitor(18e-9, "b_inp", "gnd")                            # c5
    c.resistor(10e3, "b_out", "b_inm")                            # r10
    c.capacitor(10e-9, "b_out", "b_inm")                          # c7
    c.pot(20e3, tone, "b_inp", "p2_2", "b_inm")                   # p2
    c.capacitor(27e-9, "p2_2", "c6b")                             # c6
    c.resistor(470, "c6b", "gnd")                                 # r11
    # output stage
    c.capacitor(1e-3, "b_out", "c8b")                             # c8
    c.resistor(4.7e3, "c8b", "r12b")                              # r12
    c.pot(10e3, level, "vb", "p3_2", "r12b")                      # p3
    c.resistor(22e3, "p3_2", "r20b")                              # r20
    c.capacitor(47e-9, "r20b", "q2b")                             # c9
    c.resistor(1e6, "q2b", "vb")                                  # r13
    c.npn("q2b", "vcc", "q2e", 80e-15, 80e-15, 1, 1, 500, 10)     # q2
    c.resistor(10e3, "q2e", "gnd")                                # r14
    c.resistor(1e3, "q2e", "r15b")                                # r15
    c.capacitor(1e-6, "r15b", "out")                              # c10
    c.resistor(100e3, "out", "gnd")                               # r16
    return c, "out"                                               # j2: probe out - gnd


def birdie(vol):
    """examples/birdie.jl:13-31."""
    c = Netlist()
    c.vsource("vcc", "gnd", 9.0)                                  # j3
    c.capacitor(100e-6, "gnd", "vcc")                             # c5
    c.diode("gnd", "vcc", 350e-12, 1.6)                           # d1
    c.vsource("in", "gnd", "u")                                   # j1
    c.resistor(1e6, "in", "gnd")                                  # r1
    c.capacitor(2.2e-9, "in", "base")                             # c1
    c.resistor(43e3, "base", "gnd")                               # r2
    c.resistor(430e3, "base", "vcc")                              # r3
    c.npn("base", "coll", "emit", 64.53e-15, 154.1e-15, 1.06, 1.10, 500, 12)   # t1
    c.resistor(390, "emit", "gnd")                                # r4
    c.resistor(10e3, "coll", "vcc")                               # r5
    c.capacitor(2.2e-9, "coll", "c3b")                            # c3
    c.pot(100e3, vol, "gnd", "wiper", "c3b")                      # p1
    return c, "wiper"                                             # j2


def gauss_solve(A, b):
    """Partially pivoted Gaussian elimination in the arrays' own precision (longdouble)."""
    A = A.copy()
    b = b.copy()
    n = len(b)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            b[[k, p]] = b[[p, k]]
        f = A[k + 1:, k] / A[k, k]
        A[k + 1:, k:] -= f[:, None] * A[k, k:][None, :]
        b[k + 1:] -= f * b[k]
    x = np.zeros(n, LD)
    for k in range(n - 1, -1, -1):
        x[k] = (b[k] - A[k, k + 1:] @ x[k + 1:]) / A[k, k]
    return x


def simulate(c, out_node, u, fs, tol=LD(1e-14)):
    """Unknowns: node voltages, then one branch current per voltage source and per op-amp output.
    Residual rows: KCL (currents LEAVING the node through the elements) for every node, the sources'
    and op-amps' constraint equations."""
    T = LD(1) / LD(fs)
    nn = len(c.nodes) - 1
    nb = len(c.V) + len(c.OA)
    N = nn + nb
    # constant part of the Jacobian: resistors, capacitor companions, source / op-amp stamps
    G = np.zeros((N, N), LD)

    def stamp(a, b, g):
        for (i, j, s) in ((a, a, 1), (b, b, 1), (a, b, -1), (b, a, -1)):
            if i >= 0 and j >= 0:
                G[i, j] += s * g
    for r, a, b in c.R:
        if r == 0:
            raise ValueError("a potentiometer at its end stop is a short: use an interior position")
        stamp(a, b, 1 / r)
    gc = [2 * cap / T for cap, _, _ in c.C]
    for (cap, a, b), g in zip(c.C, gc):
        stamp(a, b, g)
    for k, (p, m, _) in enumerate(c.V):
        row = nn + k
        for node, s in ((p, 1), (m, -1)):
            if node >= 0:
                G[node, row] += s          # branch current leaves `plus`, enters `minus`
                G[row, node] += s          # v(plus) - v(minus) = value
    for k, (ip, im, op, om) in enumerate(c.OA):
        row = nn + len(c.V) + k
        for node, s in ((op, 1), (om, -1)):
            if node >= 0:
                G[node, row] += s          # free output current
        for node, s in ((ip, 1), (im, -1)):
            if node >= 0:
                G[row, node] += s          # v(in+) - v(in-) = 0

    def volt(x, a):
        return x[a] if a >= 0 else LD(0)

    def assemble(x, xc, un, lam):
        """F(x) and dF/dx at one sample; lam scales every independent source and the history (source stepping)."""
        F = G @ x
        J = G.copy()
        for (cap, a, b), xk in zip(c.C, xc):
            h = lam * 2 / T * xk                       # i = (2c/T) v - (2/T) x[n-1]
            if a >= 0:
                F[a] -= h
            if b >= 0:
                F[b] += h
        for k, (_, _, val) in enumerate(c.V):
            F[nn + k] -= lam * (un if val == "u" else LD(val))

        def junction(a, b, i, g):                      # current i flows a -> b, g = di/d(va - vb)
            if a >= 0:
                F[a] += i
                J[a, a] += g
                if b >= 0:
                    J[a, b] -= g
            if b >= 0:
                F[b] -= i
                J[b, b] += g
                if a >= 0:
                    J[b, a] -= g
        for an, ca, is_, k in c.D:
            e = np.exp((volt(x, an) - volt(x, ca)) * k)
            junction(an, ca, is_ * (e - 1), is_ * k * e)
        for b, col, em, af_ise, ar_isc, ke, kc, bf, br in c.Q:
            eE = np.exp((volt(x, b) - volt(x, em)) * ke)
            eC = np.exp((volt(x, b) - volt(x, col)) * kc)
            i_f, i_r = af_ise * (eE - 1), ar_isc * (eC - 1)
            gf, gr = af_ise * ke * eE, ar_isc * kc * eC
            # port base->emitter carries i_cc + i_f/bf, port base->collector -i_cc + i_r/br, i_cc = i_f - i_r
            # written as three two-terminal junction currents:
            junction(b, em, i_f / bf, gf / bf)         # base-emitter diode
            junction(b, col, i_r / br, gr / br)        # base-collector diode
            # transport current i_cc from collector to emitter, controlled by both junction voltages
            icc = i_f - i_r
            for node, s in ((col, 1), (em, -1)):
                if node < 0:
                    continue
                F[node] += s * icc
                if b >= 0:
                    J[node, b] += s * (gf - gr)
                if em >= 0:
                    J[node, em] -= s * gf
                if col >= 0:
                    J[node, col] += s * gr
        return F, J

    junc = [(an, ca) for an, ca, _, _ in c.D] + [(b, em) for b, _, em, *_ in c.Q] + [(b, col) for b, col, *_ in c.Q]

    def newton(x, xc, un, lam):
        for it in range(400):
            F, J = assemble(x, xc, un, lam)
            dx = gauss_solve(J, -F)
            # junction-voltage limiting: no junction moves by more than 0.1 V per step
            worst = max(abs(volt(dx, a) - volt(dx, b)) for a, b in junc)
            s = min(LD(1), LD(0.1) / worst) if worst > 0 else LD(1)
            x = x + s * dx
            if s == 1 and np.max(np.abs(dx)) < tol * max(LD(1), np.max(np.abs(x))):
                # one more residual check at the accepted point
                F, _ = assemble(x, xc, un, lam)
                return x, it + 1, float(np.max(np.abs(F)))
        raise RuntimeError("Newton did not converge")

    x = np.zeros(N, LD)
    xc = [LD(0)] * len(c.C)                            # capacitor charges, x[-1] = 0 (src/ACME.jl:145)
    y = np.zeros(len(u))
    iters = 0
    worst_res = 0.0
    for n, un in enumerate(u):
        un = LD(un)
        if n == 0:                                     # power-up from zero charge: ramp the sources in
            for lam in np.linspace(0.02, 1.0, 50):
                x, it, _ = newton(x, xc, un, LD(lam))
        x, it, res = newton(x, xc, un, LD(1))
        iters += it
        worst_res = max(worst_res, res)
        y[n] = float(volt(x, c.nodes[out_node]))
        xc = [2 * cap * (volt(x, a) - volt(x, b)) - xk for (cap, a, b), xk in zip(c.C, xc)]
    return y, iters / len(u), worst_res


SUPEROVER_POTS = [(0.3, 0.5, 0.7), (0.85, 0.2, 0.4), (0.05, 0.9, 0.95), (0.6, 0.75, 0.1)]
BIRDIE = [(0.4, 0.8), (0.9, 0.25)]       # (vol, input amplitude)


def main():
    out = {}
    T = 4410
    s = np.sin(2 * np.pi * 1000 / 44100 * np.arange(T))
    ys = []
    for pots in SUPEROVER_POTS:
        c, o = superover(*pots)
        y, its, res = simulate(c, o, s, 44100)
        print(f"superover pots {pots}: {its:.2f} Newton iterations per sample, worst KCL residual {res:.1e} A, "
              f"max |y| {np.abs(y).max():.4f}", flush=True)
        ys.append(y)
    out["superover_pots"] = np.array(SUPEROVER_POTS)
    out["superover_u"] = s
    out["superover_y"] = np.array(ys)
    Tb = 4 * 4410
    sb = np.sin(2 * np.pi * 1000 / 176400 * np.arange(Tb))
    ys = []
    for vol, amp in BIRDIE:
        c, o = birdie(vol)
        y, its, res = simulate(c, o, amp * sb, 176400)
        print(f"birdie vol {vol} amplitude {amp}: {its:.2f} iterations per sample, worst residual {res:.1e} A, "
              f"max |y| {np.abs(y).max():.4f}", flush=True)
        ys.append(y)
    out["birdie_vol_amp"] = np.array(BIRDIE)
    out["birdie_u"] = sb
    out["birdie_y"] = np.array(ys)
    np.savez_compressed(os.path.join(HERE, "mna_reference.npz"), **out)
    print("wrote", os.path.join(HERE, "mna_reference.npz"))


if __name__ == "__main__":
    sys.exit(main())
