"""Tiny exact-rational dense/sparse matrix helpers used by the host derivation front end.

Matrices are lists of row lists of ``fractions.Fraction`` (the models are <= ~150x150,
seconds per derivation, run once per circuit).  ``gensolve``/``rank_factorize`` follow
src/ACME.jl:717-762 of the reference.
"""
from __future__ import annotations

from fractions import Fraction

ZERO = Fraction(0)
ONE = Fraction(1)


def shape(m, ncols=None):
    r = len(m)
    if r == 0:
        return (0, ncols or 0)
    return (r, len(m[0]))


def zeros(r, c):
    return [[ZERO] * c for _ in range(r)]


def eye(n):
    return [[ONE if i == j else ZERO for j in range(n)] for i in range(n)]


def frac_matrix(m):
    return [[v if hasattr(v, "n") and hasattr(v, "v") else Fraction(v) for v in row] for row in m]


def transpose(m, ncols=None):
    r, c = shape(m, ncols)
    return [[m[i][j] for i in range(r)] for j in range(c)]


def matmul(a, b, b_cols=None):
    """a (r x k) * b (k x c).  ``b_cols`` gives c when b has zero rows."""
    r = len(a)
    k = len(b)
    c = len(b[0]) if k else (b_cols or 0)
    out = [[ZERO] * c for _ in range(r)]
    for i in range(r):
        ai = a[i]
        oi = out[i]
        for l in range(k):
            v = ai[l]
            if v:
                bl = b[l]
                for j in range(c):
                    w = bl[j]
                    if w:
                        oi[j] += v * w
    return out


def matvec(a, x):
    return [sum((v * w for v, w in zip(row, x) if v and w), ZERO) for row in a]


def add(a, b):
    return [[x + y for x, y in zip(ra, rb)] for ra, rb in zip(a, b)]


def sub(a, b):
    return [[x - y for x, y in zip(ra, rb)] for ra, rb in zip(a, b)]


def scale(a, s):
    s = Fraction(s)
    return [[x * s for x in row] for row in a]


def hcat(*ms):
    ms = [m for m in ms]
    r = max(len(m) for m in ms)
    out = [[] for _ in range(r)]
    for m in ms:
        if len(m) == 0:
            continue
        assert len(m) == r
        for i in range(r):
            out[i].extend(m[i])
    return out


def vcat(*ms):
    out = []
    for m in ms:
        out.extend([list(row) for row in m])
    return out


def rows(m, idx):
    return [list(m[i]) for i in idx]


def cols(m, idx):
    return [[row[j] for j in idx] for row in m]


def sub_block(m, ridx, cidx):
    return [[m[i][j] for j in cidx] for i in ridx]


def is_zero(m):
    return all(not v for row in m for v in row)


def argmax_abs(m):
    """First maximal |entry| in column-major order (Julia ``argmax(abs.(m))``)."""
    r = len(m)
    c = len(m[0]) if r else 0
    best = None
    bi = bj = -1
    for j in range(c):
        for i in range(r):
            v = abs(m[i][j])
            if best is None or v > best:
                best, bi, bj = v, i, j
    return bi, bj


# eps(BigFloat) at Julia's default 256-bit precision, used in gensolve's tolerance
_EPS_BIGFLOAT = Fraction(1, 2 ** 255)


def gensolve(a, b, n, nrhs, thresh=Fraction(0.1)):
    """General solution of a*x = b (src/ACME.jl:717-747).

    ``a``: m x n, ``b``: m x nrhs dense Fraction matrices.  Returns ``(x, h)`` with x an
    n x nrhs particular solution and h an n x k basis of the homogeneous solutions, both
    dense.  Row processing order (stable sort by nnz), the 0.1 pivot threshold and the
    fewest-nonzeros column choice follow the reference.
    """
    m = len(a)
    x_cols = [dict() for _ in range(nrhs)]
    h_cols = [{j: ONE} for j in range(n)]
    if m:
        a_rows = [{j: v for j, v in enumerate(row) if v} for row in a]
        order = sorted(range(m), key=lambda i: len(a_rows[i]))
        tol = 3 * _EPS_BIGFLOAT * n
        for i in order:
            ait = a_rows[i]
            s = {}
            for j, hj in enumerate(h_cols):
                if len(ait) < len(hj):
                    v = sum((val * hj[k] for k, val in ait.items() if k in hj), ZERO)
                else:
                    v = sum((val * ait[k] for k, val in hj.items() if k in ait), ZERO)
                if v:
                    s[j] = v
            if not s:
                continue
            max_abs = max(abs(v) for v in s.values())
            if max_abs <= tol:
                continue
            lim = thresh * max_abs
            jat = [j for j in sorted(s) if abs(s[j]) >= lim]
            j = min(jat, key=lambda jj: len(h_cols[jj]))  # first minimum
            q = h_cols[j]
            sj = s[j]
            bi = b[i]
            for c in range(nrhs):
                xc = x_cols[c]
                r = bi[c] - sum((val * xc[k] for k, val in ait.items() if k in xc), ZERO)
                if r:
                    r = r / sj
                    for k, qv in q.items():
                        nv = xc.get(k, ZERO) + qv * r
                        if nv:
                            xc[k] = nv
                        else:
                            xc.pop(k, None)
            new_h = []
            for jj, hjj in enumerate(h_cols):
                if jj == j:
                    continue
                sv = s.get(jj)
                if sv:
                    f = sv / sj
                    col = dict(hjj)
                    for k, qv in q.items():
                        nv = col.get(k, ZERO) - qv * f
                        if nv:
                            col[k] = nv
                        else:
                            col.pop(k, None)
                    new_h.append(col)
                else:
                    new_h.append(hjj)
            h_cols = new_h
    x = zeros(n, nrhs)
    for c, xc in enumerate(x_cols):
        for k, v in xc.items():
            x[k][c] = v
    h = zeros(n, len(h_cols))
    for c, hc in enumerate(h_cols):
        for k, v in hc.items():
            h[k][c] = v
    return x, h


def nullspace(a, n):
    """Basis of {x : a*x = 0}; ``n`` = number of columns of a."""
    return gensolve(a, zeros(len(a), 0), n, 0)[1]


def rank_factorize(a, ncols):
    """a = c*f with f having the minimum number of rows (src/ACME.jl:749-762)."""
    nr = len(a)
    f = [list(r) for r in a]
    ns = nullspace(transpose(a, ncols), nr)      # left null space, nr x k
    c = eye(nr)
    k = len(ns[0]) if nr and ns else 0
    while k > 0:
        i, j = argmax_abs(ns)
        piv = ns[i][j]
        ci = [row[i] for row in c]
        nsj = [row[j] for row in ns]
        # c -= c[:, i] * ns[:, j]' / ns[i, j]
        for r in range(len(c)):
            if ci[r]:
                cr = c[r]
                for cc in range(len(cr)):
                    if nsj[cc]:
                        cr[cc] -= ci[r] * nsj[cc] / piv
        c = [[v for cc, v in enumerate(row) if cc != i] for row in c]
        # ns -= ns[:, j] * ns[i, :]' / ns[i, j]
        nsi = list(ns[i])
        for r in range(len(ns)):
            if nsj[r]:
                nr_ = ns[r]
                for cc in range(len(nr_)):
                    if nsi[cc]:
                        nr_[cc] -= nsj[r] * nsi[cc] / piv
        ns = [[v for cc, v in enumerate(row) if cc != j] for r, row in enumerate(ns) if r != i]
        f = [row for r, row in enumerate(f) if r != i]
        k -= 1
    return c, f


def to_float(m):
    return [[float(v) for v in row] for row in m]
