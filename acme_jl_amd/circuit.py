"""Circuit description: elements, netlist, incidence and topology matrices.

Host-side (CPU, exact rational) mirror of the reference's schematic front end.  It
exists only because the hot path's *inputs* (the DiscreteModel matrices) are produced
by this front end and no Julia is available next to the GPU; it is NOT part of the
GPU hot path.

Reference interfaces mirrored here (file:line relative to the ACME.jl tree):
  * Element + prepare_element_matrices ........ src/ACME.jl:21-112
  * element constructors ...................... src/elements.jl:16-551
  * Circuit / add! / connect! / disconnect! ... src/circuit.jl:22-206
  * incidence, topomat! ....................... src/circuit.jl:51-66, 208-252
  * nonlinear_eq_func (element table order) ... src/circuit.jl:68-86

All matrix entries are ``fractions.Fraction`` holding the *binary* value of the given
Python number, exactly like ``convert(Rational{BigInt}, ::Float64)`` at
src/circuit.jl:43-45.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from fractions import Fraction

from .batchval import BVal

# ---------------------------------------------------------------------------------
# nonlinear element kinds: shared numbering with oracle/acme_ref.h and include/acme_hip.h
# ---------------------------------------------------------------------------------
KIND_NONE = 0
KIND_DIODE = 1       # par = [is, eta]                                  src/elements.jl:236-245
KIND_BJT = 2         # par = [ise, isc, etae, etac, bf, br, ile, ilc,
                     #        etael, etacl, vaf, var, ikf, ikr]         src/elements.jl:309-406
KIND_POT = 3         # par = [r]                                        src/elements.jl:20-31
KIND_MOSFET = 4      # par = [polarity, lambda, nvt, vt0..vt3, nalpha, a0..a3]  src/elements.jl:436-481
KIND_MACAK = 5       # par = [gain, scale]                              src/elements.jl:536-551
KIND_JA = 6          # par = [Ms, a, alpha, c, k]                       src/elements.jl:100-135
MAX_ELEM_PAR = 16

# (nq, nn) of each kind's nonlinear function
KIND_SHAPE = {
    KIND_DIODE: (2, 1),
    KIND_BJT: (4, 2),
    KIND_POT: (5, 2),
    KIND_MOSFET: (3, 1),
    KIND_MACAK: (2, 1),
    KIND_JA: (4, 1),
}


def _frac(v):
    if isinstance(v, (Fraction, BVal)):   # BVal: a batch of instances (montecarlo.derive_batch)
        return v
    if isinstance(v, bool):
        return Fraction(int(v))
    if isinstance(v, int):
        return Fraction(v)
    if isinstance(v, float):
        if math.isinf(v) or math.isnan(v):
            raise ValueError("non-finite matrix entry")
        return Fraction(v)  # exact binary value
    return Fraction(v)


def _as_matrix(m):
    """Julia ``hcat(x)``: scalar -> 1x1, vector -> n x 1 column, matrix unchanged."""
    if isinstance(m, (int, float, Fraction, BVal)):
        return [[_frac(m)]]
    m = list(m)
    if len(m) == 0:
        return []
    if isinstance(m[0], (list, tuple)):
        return [[_frac(v) for v in row] for row in m]
    return [[_frac(v)] for v in m]


_MAT_DIMS = OrderedDict([
    ("mv", ("nl", "nb")), ("mi", ("nl", "nb")), ("mx", ("nl", "nx")),
    ("mxd", ("nl", "nx")), ("mq", ("nl", "nq")), ("mu", ("nl", "nu")),
    ("u0", ("nl", "n0")),
    ("pv", ("ny", "nb")), ("pi", ("ny", "nb")), ("px", ("ny", "nx")),
    ("pxd", ("ny", "nx")), ("pq", ("ny", "nq")),
])


class Element:
    """A circuit element: 12 stamp matrices + optional nonlinear descriptor + pins.

    Mirrors ``struct Element`` (src/ACME.jl:58-98).  ``nonlinear`` is a list of
    ``(kind, params)`` descriptors instead of a closure, because closures cannot cross
    the C ABI; the list has more than one entry only for composite elements.
    """

    def __init__(self, nonlinear=None, ports=None, pins=None, **mats):
        matrices = {}
        sizes = {"n0": 1}
        for name, m in mats.items():
            if name not in _MAT_DIMS:
                raise TypeError(f"unknown element matrix {name}")
            if m is None:
                continue
            mm = _as_matrix(m)
            nr = len(mm)
            nc = len(mm[0]) if nr else 0
            # zero-row matrices given as nested empty lists keep their column count 0
            for sym, s in zip(_MAT_DIMS[name], (nr, nc)):
                if sizes.setdefault(sym, s) != s:
                    raise ValueError(f"Inconsistent sizes for {sym}")
            matrices[name] = mm
        for name, (rs, cs) in _MAT_DIMS.items():
            if name not in matrices:
                nr = sizes.setdefault(rs, 0)
                nc = sizes.setdefault(cs, 0)
                matrices[name] = [[Fraction(0)] * nc for _ in range(nr)]
        self.m = matrices
        self.sizes = sizes
        self.nonlinear = list(nonlinear) if nonlinear else []
        if ports is not None:
            pins = OrderedDict()
            for branch, (p, n) in enumerate(ports, start=1):
                pins.setdefault(str(p), []).append((branch, 1))
                pins.setdefault(str(n), []).append((branch, -1))
        if pins is None:
            pins = OrderedDict()
            for i in range(1, 2 * sizes["nb"] + 1):
                pins[str(i)] = [((i + 1) // 2, 2 * (i % 2) - 1)]
        self.pins = pins

    # size accessors (src/ACME.jl:105-110)
    @property
    def nb(self): return self.sizes["nb"]
    @property
    def nx(self): return self.sizes["nx"]
    @property
    def nq(self): return self.sizes["nq"]
    @property
    def nu(self): return self.sizes["nu"]
    @property
    def nl(self): return self.sizes["nl"]
    @property
    def ny(self): return self.sizes["ny"]
    @property
    def nn(self): return self.nb + self.nx + self.nq - self.nl


def _eye(n, s=1):
    return [[s if i == j else 0 for j in range(n)] for i in range(n)]


def _zeros(r, c):
    return [[0] * c for _ in range(r)]


# ---------------------------------------------------------------------------------
# element constructors (src/elements.jl)
# ---------------------------------------------------------------------------------
def resistor(r):
    """src/elements.jl:16"""
    return Element(mv=-1, mi=r)


def potentiometer(r, pos=None):
    """src/elements.jl:18-31 (fixed position / position as extra input)."""
    if pos is not None:
        return Element(mv=_eye(2, -1), mi=[[r * pos, 0], [0, r * (1 - pos)]],
                       ports=[(1, 2), (2, 3)])
    scale = 1
    if isinstance(r, BVal):
        # a batch of instances shares ONE element table: the table holds the structure instance's
        # track resistance r_ref and the q rows carrying the currents are scaled by r / r_ref, so
        # that  v - r_ref*pos*(i*r/r_ref) = v - r*pos*i  holds in every instance
        r_ref = float(r)
        scale = r / Fraction(r_ref)
        r = r_ref
    return Element(
        mv=_eye(2) + _zeros(3, 2),
        mi=_zeros(2, 2) + [[scale, 0], [0, scale]] + _zeros(1, 2),
        mq=_eye(5, -1), mu=[0, 0, 0, 0, -1],
        nonlinear=[(KIND_POT, [float(r)])],
        ports=[(1, 2), (2, 3)])


def capacitor(c):
    """src/elements.jl:40"""
    return Element(mv=[c, 0], mi=[0, 1], mx=[-1, 0], mxd=[0, -1])


def inductor(l=None, *, ja=False, n=230, **kw):
    """src/elements.jl:49 and the Jiles-Atherton form :167-168."""
    if ja:
        return transformer_ja(ns=[n], **kw)
    return Element(mv=[1, 0], mi=[0, l], mx=[0, -1], mxd=[-1, 0])


def transformer(l1, l2, coupling_coefficient=1, mutual_coupling=None):
    """src/elements.jl:63-68"""
    if mutual_coupling is None:
        mutual_coupling = coupling_coefficient * math.sqrt(l1 * l2)
    return Element(mv=[[1, 0], [0, 1], [0, 0], [0, 0]],
                   mi=[[0, 0], [0, 0], [l1, mutual_coupling], [mutual_coupling, l2]],
                   mx=[[0, 0], [0, 0], [-1, 0], [0, -1]],
                   mxd=[[-1, 0], [0, -1], [0, 0], [0, 0]],
                   ports=[("primary1", "primary2"), ("secondary1", "secondary2")])


def transformer_ja(ns=(), D=2.4e-2, A=4.54e-5, a=14.1, alpha=5e-5, c=0.55, k=17.8,
                   Ms=2.75e5):
    """Jiles-Atherton transformer, src/elements.jl:100-135."""
    mu0 = 1.2566370614e-6
    ns = list(ns)
    nw = len(ns)
    mv = [[1 if i == j else 0 for j in range(nw)] for i in range(nw + 5)]
    mi = _zeros(nw, nw) + [list(ns)] + _zeros(4, nw)
    mx = _zeros(nw, 2) + [[-math.pi * D, 0], [-1 / a, -alpha / a], [0, -1], [0, 0], [0, 0]]
    mxd = [[-mu0 * A * n_, -mu0 * n_ * A] for n_ in ns] + \
        [[0, 0], [0, 0], [0, 0], [-1, 0], [0, -1]]
    mq = _zeros(nw + 1, 4) + _eye(4)
    return Element(mv=mv, mi=mi, mx=mx, mxd=mxd, mq=mq,
                   nonlinear=[(KIND_JA, [float(Ms), float(a), float(alpha), float(c), float(k)])])


def voltagesource(v=None, rs=0):
    """src/elements.jl:181-183"""
    if v is None:
        return Element(mv=1, mi=-rs, mu=1, ports=[("+", "-")])
    return Element(mv=1, mi=-rs, u0=v, ports=[("+", "-")])


def currentsource(i=None, gp=0):
    """src/elements.jl:197-199"""
    if i is None:
        return Element(mv=gp, mi=-1, mu=1, ports=[("+", "-")])
    return Element(mv=gp, mi=-1, u0=i, ports=[("+", "-")])


def voltageprobe(gp=0):
    """src/elements.jl:210-211"""
    return Element(mv=-gp, mi=1, pv=1, ports=[("+", "-")])


def currentprobe(rs=0):
    """src/elements.jl:223-224"""
    return Element(mv=1, mi=-rs, pi=1, ports=[("+", "-")])


def diode(is_=1e-12, eta=1):
    """src/elements.jl:235-245"""
    return Element(mv=[1, 0], mi=[0, 1], mq=[[-1, 0], [0, -1]], ports=[("+", "-")],
                   nonlinear=[(KIND_DIODE, [float(is_), float(eta)])])


def bjt(typ, is_=1e-12, eta=1, isc=None, ise=None, etac=None, etae=None, bf=1000, br=10,
        ile=0, ilc=0, etacl=None, etael=None, vaf=math.inf, var=math.inf,
        ikf=math.inf, ikr=math.inf, re=0, rc=0, rb=0):
    """Gummel-Poon / Ebers-Moll BJT, src/elements.jl:309-406."""
    isc = is_ if isc is None else isc
    ise = is_ if ise is None else ise
    etac = eta if etac is None else etac
    etae = eta if etae is None else etae
    etacl = etac if etacl is None else etacl
    etael = etae if etael is None else etael
    if typ == "npn":
        polarity = 1
    elif typ == "pnp":
        polarity = -1
    else:
        raise ValueError(f"Unknown bjt type {typ}, must be npn or pnp")
    par = [ise, isc, etae, etac, bf, br, ile, ilc, etael, etacl, vaf, var, ikf, ikr]
    return Element(mv=[[1, 0], [0, 1], [0, 0], [0, 0]],
                   mi=[[-(re + rb), -rb], [-rb, -(rc + rb)], [1, 0], [0, 1]],
                   mq=_eye(4, -polarity),
                   nonlinear=[(KIND_BJT, [float(p) for p in par])],
                   ports=[("base", "emitter"), ("base", "collector")])


def mosfet(typ, vt=0.7, alpha=2e-5, lam=0):
    """src/elements.jl:436-481 (polynomial vt/alpha with up to 4 coefficients each)."""
    if typ == "n":
        polarity = 1
    elif typ == "p":
        polarity = -1
    else:
        raise ValueError(f"Unknown mosfet type {typ}, must be n or p")
    vt = tuple(vt) if isinstance(vt, (tuple, list)) else (vt,)
    alpha = tuple(alpha) if isinstance(alpha, (tuple, list)) else (alpha,)
    if len(vt) > 4 or len(alpha) > 4:
        raise ValueError("at most 4 polynomial coefficients supported")
    par = [float(polarity), float(lam), float(len(vt))] + \
        [float(v) for v in vt] + [0.0] * (4 - len(vt)) + \
        [float(len(alpha))] + [float(a) for a in alpha] + [0.0] * (4 - len(alpha))
    p = polarity
    return Element(mv=[[-1, 0], [0, -1], [0, 0], [0, 0]],
                   mi=[[0, 0], [0, 0], [0, -1], [1, 0]],
                   mq=[[p, 0, 0], [0, p, 0], [0, 0, p], [0, 0, 0]],
                   ports=[("gate", "source"), ("drain", "source")],
                   nonlinear=[(KIND_MOSFET, par)])


def opamp(maxgain=math.inf, gain_bw_prod=math.inf):
    """Linear op-amp, src/elements.jl:508-517."""
    ports = [("in+", "in-"), ("out+", "out-")]
    if math.isinf(gain_bw_prod):
        return Element(mv=[[0, 0], [1, -1 / maxgain]], mi=[[1, 0], [0, 0]], ports=ports)
    return Element(mv=[[0, 0], [-1 / math.sqrt(1 - 1 / maxgain ** 2), 0], [0, -1]],
                   mi=[[1, 0], [0, 0], [0, 0]],
                   mx=[0, (0.0 if math.isinf(maxgain) else 1 / math.sqrt(maxgain ** 2 - 1)), 1],
                   mxd=[0, 1 / (2 * math.pi * gain_bw_prod), 0], ports=ports)


def opamp_macak(gain, vomin, vomax):
    """Clipping (tanh) op-amp, src/elements.jl:536-551."""
    offset = 0.5 * (vomin + vomax)
    scale = 0.5 * (vomax - vomin)
    return Element(mv=[[0, 0], [1, 0], [0, 1]], mi=[[1, 0], [0, 0], [0, 0]],
                   mq=[[0, 0], [-1, 0], [0, -1]], u0=[0, 0, offset],
                   nonlinear=[(KIND_MACAK, [float(gain), float(scale)])],
                   ports=[("in+", "in-"), ("out+", "out-")])


# ---------------------------------------------------------------------------------
# Circuit (src/circuit.jl:22-206)
# ---------------------------------------------------------------------------------
class Circuit:
    def __init__(self):
        self.elements = OrderedDict()
        self.nets = []          # list of nets; a net is a list of (designator, pin)
        self.net_names = {}     # name -> net (object identity)
        self._gensym = 0

    # --- construction -------------------------------------------------------------
    def add(self, designator, elem=None):
        """add!(c, designator, elem) / add!(c, elem) (src/circuit.jl:94-117)."""
        if elem is None:
            elem = designator
            self._gensym += 1
            designator = f"##{self._gensym}"
        if designator in self.elements:
            self.delete(designator)
        for pin in elem.pins:
            self.nets.append([(designator, pin)])
        self.elements[designator] = elem
        return designator

    def delete(self, designator):
        """delete!(c, designator) (src/circuit.jl:125-130)."""
        for net in self.nets:
            net[:] = [ep for ep in net if ep[0] != designator]
        del self.elements[designator]

    def _netfor(self, p):
        if isinstance(p, tuple):
            key = (p[0], str(p[1]))
            for net in self.nets:
                if key in net:
                    return net
            raise ValueError(f"Unknown pin {p}")
        if p not in self.net_names:
            net = []
            self.net_names[p] = net
            self.nets.append(net)
        return self.net_names[p]

    def connect(self, *pins):
        """connect!(c, pins...) (src/circuit.jl:175-188).  A pin is ``(designator, pin)``,
        a bare string is a named net."""
        nets = []
        for p in pins:
            n = self._netfor(p)
            if not any(n is m for m in nets):
                nets.append(n)
        for net in nets[1:]:
            nets[0].extend(net)
            idx = next(i for i, m in enumerate(self.nets) if m is net)
            del self.nets[idx]
            for name, named in list(self.net_names.items()):
                if named is net:
                    self.net_names[name] = nets[0]

    def disconnect(self, pin):
        """disconnect!(c, pin) (src/circuit.jl:190-206)."""
        pin = (pin[0], str(pin[1]))
        net = self._netfor(pin)
        net[:] = [p for p in net if p != pin]
        self.nets.append([pin])

    # --- aggregate sizes (src/circuit.jl:33-35) --------------------------------------
    def _sum(self, attr):
        return sum(getattr(e, attr) for e in self.elements.values())

    @property
    def nb(self): return self._sum("nb")
    @property
    def nx(self): return self._sum("nx")
    @property
    def nq(self): return self._sum("nq")
    @property
    def nu(self): return self._sum("nu")
    @property
    def nl(self): return self._sum("nl")
    @property
    def ny(self): return self._sum("ny")
    @property
    def nn(self): return self._sum("nn")

    def blockdiag(self, name):
        """Block-diagonal concatenation of one stamp matrix over all elements
        (src/circuit.jl:37-47); returns a dense list-of-lists of Fractions."""
        rs, cs = _MAT_DIMS[name]
        rows = sum(e.sizes[rs] for e in self.elements.values())
        cols = sum(e.sizes[cs] for e in self.elements.values())
        out = [[Fraction(0)] * cols for _ in range(rows)]
        r0 = c0 = 0
        for e in self.elements.values():
            m = e.m[name]
            for i, row in enumerate(m):
                for j, v in enumerate(row):
                    out[r0 + i][c0 + j] = v
            r0 += e.sizes[rs]
            c0 += e.sizes[cs]
        return out

    def u0(self):
        """vcat of element u0 columns (src/circuit.jl:49)."""
        out = []
        for e in self.elements.values():
            out.extend([row[0]] for row in e.m["u0"])
        return out

    def branch_offset(self, designator):
        off = 0
        for des, el in self.elements.items():
            if des == designator:
                return off
            off += el.nb
        raise ValueError("Element not found in circuit")

    def incidence(self):
        """Node-branch incidence matrix (src/circuit.jl:51-66), dense ints."""
        nb = self.nb
        inc = [[0] * nb for _ in self.nets]
        for row, net in enumerate(self.nets):
            for elemname, pinname in net:
                off = self.branch_offset(elemname)
                for branch, polarity in self.elements[elemname].pins[pinname]:
                    inc[row][off + branch - 1] += polarity
        return inc

    def topomat(self):
        return topomat(self.incidence())

    def nonlinear_table(self, elem_idxs=None):
        """Element-descriptor table in CircuitNLFunc order (src/circuit.jl:68-86):
        list of dicts {kind, par, nq, nn, qoff, roff}; elements with nn == nq == 0 are
        skipped, q offsets are cumulative over the *selected* elements."""
        elems = list(self.elements.values())
        if elem_idxs is not None:
            elems = [elems[i] for i in elem_idxs]
        table = []
        qoff = roff = 0
        for e in elems:
            if e.nn == 0 and e.nq == 0:
                continue
            if not e.nonlinear:
                if e.nq:
                    raise ValueError("element has nq > 0 but no nonlinear descriptor")
                continue
            for kind, par in e.nonlinear:
                knq, knn = KIND_SHAPE[kind]
                table.append(dict(kind=kind, par=list(par), nq=knq, nn=knn,
                                  qoff=qoff, roff=roff))
                qoff += knq
                roff += knn
        return table


def topomat(incidence):
    """Fundamental cut-set / loop matrices from the incidence matrix
    (``topomat!``, src/circuit.jl:208-249).  Returns (tv, ti) as dense int lists."""
    inc = [list(r) for r in incidence]
    nrows = len(inc)
    ncols = len(inc[0]) if nrows else 0
    for r in inc:
        for v in r:
            assert v in (-1, 0, 1)
    for c in range(ncols):
        assert sum(inc[r][c] for r in range(nrows)) == 0
    t = [False] * ncols
    row = 0
    for col in range(ncols):
        rows = [r for r in range(row, nrows) if inc[r][col] != 0]
        assert len(rows) <= 2
        if not rows:
            continue
        t[col] = True
        if rows[0] != row:
            inc[rows[0]], inc[row] = inc[row], inc[rows[0]]
        if len(rows) == 2:
            assert inc[row][col] + inc[rows[1]][col] == 0
            inc[rows[1]] = [a + b for a, b in zip(inc[rows[1]], inc[row])]
        if inc[row][col] < 0:
            inc[row] = [-a for a in inc[row]]
        for r in range(row):
            if inc[r][col] == 1:
                inc[r] = [a - b for a, b in zip(inc[r], inc[row])]
            elif inc[r][col] == -1:
                inc[r] = [a + b for a, b in zip(inc[r], inc[row])]
        row += 1
    ti = [inc[r] for r in range(row)]
    tcols = [c for c in range(ncols) if t[c]]
    lcols = [c for c in range(ncols) if not t[c]]
    nlk = len(lcols)
    tv = [[0] * ncols for _ in range(nlk)]
    for i in range(nlk):
        for jj, c in enumerate(tcols):
            tv[i][c] = -ti[jj][lcols[i]]      # -dl'
        tv[i][lcols[i]] = 1
    return tv, ti
