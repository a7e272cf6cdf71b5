"""``BVal``: one exact rational number that decides, N float64 numbers that follow.

The exact-rational derivation (``derive.py`` / ``ratmat.py``, after src/ACME.jl:264-451,717-777)
takes every structural decision -- which entries are zero, which pivot ``gensolve`` picks, the
rank of a matrix -- from exact arithmetic.  For a Monte-Carlo sweep over component values the
circuit topology, hence (generically) all of those decisions, are the same for every instance;
only the numbers differ.  A ``BVal`` carries

* ``n``: a ``Fraction`` -- the value in one *structure instance* with generic component values;
  truthiness, ``abs``/comparisons and ``float()`` look at this number only, so the derivation
  takes exactly the path it takes for that instance, and
* ``v``: the value in each of the N instances -- a double-double array (``DD``) updated by the
  same arithmetic.

Running the unmodified derivation code on matrices of ``BVal`` therefore *replays the structure*
of one exact derivation on N instances at once; an entry whose exact value
cancels to zero is dropped together with its rounding noise.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np


_SPLIT = 134217729.0   # 2^27 + 1 (Veltkamp split)


def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _two_prod(a, b):
    p = a * b
    ca = _SPLIT * a
    ah = ca - (ca - a)
    al = a - ah
    cb = _SPLIT * b
    bh = cb - (cb - b)
    bl = b - bh
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


class DD:
    """Double-double numbers (hi + lo, ~32 significant digits) on numpy arrays or scalars.

    The exact derivation only rounds once, at the end.  Replaying it in plain float64 loses up to
    1e-6 relative accuracy in the varying-pot superover's matrices (component admittances span
    nine decades, gensolve / rank_factorize cancel heavily) -- enough to change outputs by 1 %.
    With double-double the replay error stays far below the final float64 rounding."""
    __slots__ = ("h", "l")

    def __init__(self, h, l=0.0):
        self.h = h
        self.l = l

    @staticmethod
    def of(x):
        if isinstance(x, DD):
            return x
        if isinstance(x, Fraction):
            h = float(x)
            return DD(h, float(x - Fraction(h)))
        return DD(float(x) if not isinstance(x, np.ndarray) else x, 0.0)

    def _norm(s, e):
        h = s + e
        return DD(h, e - (h - s))
    _norm = staticmethod(_norm)

    def __add__(self, o):
        o = DD.of(o)
        s, e = _two_sum(self.h, o.h)
        return DD._norm(s, e + (self.l + o.l))

    def __sub__(self, o):
        o = DD.of(o)
        s, e = _two_sum(self.h, -o.h)
        return DD._norm(s, e + (self.l - o.l))

    def __neg__(self):
        return DD(-self.h, -self.l)

    def __mul__(self, o):
        o = DD.of(o)
        p, e = _two_prod(self.h, o.h)
        return DD._norm(p, e + (self.h * o.l + self.l * o.h))

    def __truediv__(self, o):
        o = DD.of(o)
        q1 = self.h / o.h
        r = self - o * DD(q1)
        q2 = r.h / o.h
        r = r - o * DD(q2)
        q3 = r.h / o.h
        s, e = _two_sum(q1, q2)
        return DD._norm(s, e + q3)

    def __abs__(self):
        neg = self.h < 0
        return DD(np.where(neg, -self.h, self.h), np.where(neg, -self.l, self.l))

    def to_float(self):
        return self.h + self.l


def _split(o):
    if isinstance(o, BVal):
        return o.n, o.v
    if isinstance(o, (int, Fraction)):
        f = Fraction(o)
        return f, DD.of(f)
    if isinstance(o, float):
        return Fraction(o), DD(o)
    return None, None


class BVal:
    __slots__ = ("n", "v")

    def __init__(self, n, v):
        self.n = n
        self.v = v

    # --- arithmetic ---------------------------------------------------------------------
    def __add__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(self.n + n, self.v + v)
    __radd__ = __add__

    def __sub__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(self.n - n, self.v - v)

    def __rsub__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(n - self.n, v - self.v)

    def __mul__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(self.n * n, self.v * v)
    __rmul__ = __mul__

    def __truediv__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(self.n / n, self.v / v)

    def __rtruediv__(self, o):
        n, v = _split(o)
        return NotImplemented if n is None else BVal(n / self.n, v / self.v)

    def __neg__(self):
        return BVal(-self.n, -self.v)

    def __pos__(self):
        return self

    def __abs__(self):
        return BVal(abs(self.n), abs(self.v))

    # --- decisions: the structure instance only --------------------------------------------
    def __bool__(self):
        return self.n != 0

    def __float__(self):
        return float(self.n)

    def _cmp(self, o):
        n, _ = _split(o)
        return n

    def __lt__(self, o): return self.n < self._cmp(o)
    def __le__(self, o): return self.n <= self._cmp(o)
    def __gt__(self, o): return self.n > self._cmp(o)
    def __ge__(self, o): return self.n >= self._cmp(o)
    def __eq__(self, o): return self.n == self._cmp(o)
    def __ne__(self, o): return self.n != self._cmp(o)
    __hash__ = None

    def __repr__(self):
        return f"BVal({float(self.n):.6g}; N={np.size(self.v.h)})"


def values(x, n):
    """float64 array [n] of the per-instance values of a BVal or an exact constant."""
    if isinstance(x, BVal):
        return np.broadcast_to(np.asarray(x.v.to_float(), dtype=np.float64), (n,))
    return np.full(n, float(x))


def structure(x):
    """float of the structure instance's value."""
    return float(x.n) if isinstance(x, BVal) else float(x)
