"""Fast per-instance front end for component-tolerance sweeps (BASELINE
config 4): N model blocks of one circuit topology from ONE pass through the derivation.

The exact-rational derivation (``derive.py``; src/ACME.jl:150-464,717-777) costs ~0.2 s per
superover model in pure Python -- hours for the 65 536 instances of config 4.  Here the same,
unmodified derivation code runs once on matrices of ``batchval.BVal``: every structural decision
(zero pattern, ``gensolve`` pivots, ranks, the nonlinearity decomposition, ``reduce_pdims``) is
taken from an exact *structure instance* with generic component values, while N float64 instances
ride along through the same arithmetic.  All instances therefore share dimensions, element table
and the bases of ``z``/``p`` -- what ``acme_batch_set_matrices`` requires -- and their matrices
agree with an exact per-instance derivation to rounding (checked in
``tests/test_montecarlo.py``; outputs are basis independent).

The construction-time solves (``initial_solution``, src/ACME.jl:453-464, and the folding of
constant sub-problems, :196-228) are done for all instances at once by a numpy-batched restatement
of the reference's homotopy/Newton iteration (each instance follows the reference's own path:
those solutions are only defined up to the 1e-10 residual tolerance, and e.g. superover's folded
reverse-biased diode makes the outputs sensitive to them at the 1e-6 level).

Host-side and one-off per sweep, like the rest of the front end: no GPU involved, the oracle is
not used.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np

from . import derive as D
from . import ratmat as rm
from .batchval import BVal, DD, values
from .circuit import KIND_BJT, KIND_DIODE, KIND_POT
from .hostsolve import eval_table
from .model import CachingHomotopySolver, DiscreteModel, HomotopySolver


# ---------------------------------------------------------------------------------------------
# numpy-batched element functions and Newton (construction-time solves only)
# ---------------------------------------------------------------------------------------------
def _eval_table_batch(table, q, nn, nq):
    """q: [N, nq] -> res [N, nn], Jq [N, nn, nq]; same formulas as hostsolve.eval_element."""
    n = q.shape[0]
    res = np.zeros((n, nn))
    jq = np.zeros((n, nn, nq))
    for e in table:
        k, par, qo, ro = e["kind"], e["par"], e["qoff"], e["roff"]
        if k == KIND_DIODE:
            is_, eta = par[0], par[1]
            ex = np.exp(q[:, qo] * (1 / (25e-3 * eta)))
            res[:, ro] = is_ * (ex - 1) - q[:, qo + 1]
            jq[:, ro, qo] = is_ / (25e-3 * eta) * ex
            jq[:, ro, qo + 1] = -1.0
        elif k == KIND_POT:
            r = par[0]
            v1, v2, i1, i2, pos = (q[:, qo + j] for j in range(5))
            res[:, ro] = v1 - r * pos * i1
            res[:, ro + 1] = v2 - r * (1 - pos) * i2
            jq[:, ro, qo] = 1.0
            jq[:, ro, qo + 2] = -r * pos
            jq[:, ro, qo + 4] = -r * i1
            jq[:, ro + 1, qo + 1] = 1.0
            jq[:, ro + 1, qo + 3] = -r * (1 - pos)
            jq[:, ro + 1, qo + 4] = -r * i2
        elif k == KIND_BJT and all(np.isinf(par[10:14])) and par[6] == 0 and par[7] == 0:
            ise, isc, etae, etac, bf, br = par[:6]
            vE, vC, iE, iC = (q[:, qo + j] for j in range(4))
            expE = np.exp(vE * (1 / (25e-3 * etae)))
            expC = np.exp(vC * (1 / (25e-3 * etac)))
            i_f = (bf / (1 + bf) * ise) * (expE - 1)
            i_r = (br / (1 + br) * isc) * (expC - 1)
            di_f1 = (bf / (1 + bf) * ise / (25e-3 * etae)) * expE
            di_r2 = (br / (1 + br) * isc / (25e-3 * etac)) * expC
            i_cc = i_f - i_r
            res[:, ro] = i_cc + (1 / bf) * i_f - iE
            res[:, ro + 1] = -i_cc + (1 / br) * i_r - iC
            jq[:, ro, qo] = di_f1 + (1 / bf) * di_f1
            jq[:, ro, qo + 1] = -di_r2
            jq[:, ro, qo + 2] = -1.0
            jq[:, ro + 1, qo] = -di_f1
            jq[:, ro + 1, qo + 1] = di_r2 + (1 / br) * di_r2
            jq[:, ro + 1, qo + 3] = -1.0
        else:   # kinds without a vectorised form: instance by instance through the scalar code
            for i in range(n):
                r_, j_ = eval_table([e], [0.0] * qo + list(q[i, qo:qo + e["nq"]]), ro + e["nn"], qo + e["nq"])
                res[i, ro:ro + e["nn"]] = r_[ro:ro + e["nn"]]
                jq[i, ro:ro + e["nn"], qo:qo + e["nq"]] = np.array(j_)[ro:ro + e["nn"], qo:qo + e["nq"]]
    return res, jq


class _BatchHomotopy:
    """HomotopySolver{SimpleSolver} on ParametricNonLinEq(f, nn, nq) with p == q
    (src/solvers.jl:23-28,151-302), for N equations at once: the same per-instance state machine
    as the GPU kernel, in numpy.  Used for the construction-time solves only."""

    def __init__(self, table, fq, nn, tol=1e-10, maxiter=500):
        self.table, self.fq, self.nn, self.tol, self.maxiter = table, fq, nn, tol, maxiter
        self.n, self.nq = fq.shape[0], fq.shape[1]
        z0 = np.zeros((self.n, nn))
        self.last_p = np.zeros((self.n, self.nq))
        self.last_z = z0
        _, j, jq = self._evaluate(np.arange(self.n), self.last_p, z0)
        self.last_jinvjp = np.linalg.solve(j, jq)               # J \ Jp with Jp = Jq (p == q)

    def _evaluate(self, idx, p, z):
        q = p + np.einsum("nqj,nj->nq", self.fq[idx], z)
        res, jq = _eval_table_batch(self.table, q, self.nn, self.nq)
        return res, np.einsum("nrq,nqj->nrj", jq, self.fq[idx]), jq

    def _base(self, idx, p):
        """solve(::SimpleSolver, p) for the instances idx; returns (z, converged)."""
        z = self.last_z[idx] - np.einsum("nrq,nq->nr", self.last_jinvjp[idx], p - self.last_p[idx])
        conv = np.zeros(idx.size, dtype=bool)
        act = np.ones(idx.size, dtype=bool)
        jq_acc = np.zeros((idx.size, self.nn, self.nq))
        j_acc = np.zeros((idx.size, self.nn, self.nn))
        for _ in range(self.maxiter):
            w = np.nonzero(act)[0]
            if w.size == 0:
                break
            res, j, jq = self._evaluate(idx[w], p[w], z[w])
            rmax = np.abs(res).max(axis=1)
            finite = np.isfinite(rmax) & np.isfinite(j).all(axis=(1, 2))
            with np.errstate(all="ignore"):
                sing = finite & (np.abs(np.linalg.det(np.where(finite[:, None, None], j, np.eye(self.nn)))) == 0.0)
            ok = finite & ~sing
            done = ok & (rmax < self.tol)
            conv[w[done]] = True
            jq_acc[w[done]], j_acc[w[done]] = jq[done], j[done]
            step = ok & ~done
            if step.any():
                z[w[step]] -= np.linalg.solve(j[step], res[step][:, :, None])[:, :, 0]
            act[w[~step]] = False
        c = np.nonzero(conv)[0]
        if c.size:                                               # accepted: new extrapolation origin
            self.last_p[idx[c]] = p[c]
            self.last_z[idx[c]] = z[c]
            self.last_jinvjp[idx[c]] = np.linalg.solve(j_acc[c], jq_acc[c])
        return z, conv

    def solve(self, p):
        n = self.n
        allidx = np.arange(n)
        z, conv = self._base(allidx, p)
        need = ~conv
        a = np.full(n, 0.5)
        best = np.zeros(n)
        start = self.last_p.copy()
        while need.any():
            idx = np.nonzero(need)[0]
            pa = start[idx] * (1 - a[idx])[:, None] + a[idx][:, None] * p[idx]
            zi, ci = self._base(idx, pa)
            z[idx] = zi
            conv[idx] = ci
            good = idx[ci]
            best[good] = a[good]
            a[good] = 1.0
            bad = idx[~ci]
            new_a = (a[bad] + best[bad]) / 2
            stuck = ~((best[bad] < new_a) & (new_a < a[bad]))
            a[bad] = new_a
            need[bad[stuck]] = False
            need[good[best[good] >= 1.0]] = False
        return z, conv


def _initial_solution_batch(table, fq, q0, nn, device=None):
    """initial_solution (src/ACME.jl:453-464) for N instances: homotopy on q from 0 to q0_i with z
    starting at 0, each instance following the reference's iteration (the result is only defined
    up to the solver tolerance, so the path matters for parity).  fq: [N, nq, nn], q0: [N, nq].

    ``device``: None -> the numpy restatement above (no GPU needed); else a dict(lib=..., device=...)
    -> all N solves in one batched ``acme_batch_solve`` on the GPU (``analysis.solve_rays``: the
    same homotopy path, walked by the same kernel that later runs the models).
    (Models whose private blocks do not fit the LDS run in the LOW-LDS kernels: nothing is refused.)"""
    if nn == 0:
        return np.zeros((q0.shape[0], 0))
    if device is not None:
        from .analysis import solve_rays
        z, conv = solve_rays(table, None, nn, q0.shape[1], q0, fq, lib=device.get("lib"), device=device.get("device"))
        if not conv.all():
            raise RuntimeError("Failed to find initial solution")
        device["solved"] = device.get("solved", 0) + q0.shape[0]
        return z
    z, conv = _BatchHomotopy(table, fq, nn).solve(q0)
    if not conv.all():
        raise RuntimeError("Failed to find initial solution")
    return z


# ---------------------------------------------------------------------------------------------
# the batch derivation
# ---------------------------------------------------------------------------------------------
def _arr(m, n, shape):
    """[N, *shape] float64 array of the instances of a (possibly empty) matrix of BVal/Fraction."""
    out = np.zeros((n,) + shape)
    if out.size == 0:
        return out
    if len(shape) == 1:
        for i, v in enumerate(m):
            out[:, i] = values(v, n)
    else:
        for i, row in enumerate(m):
            for j, v in enumerate(row):
                if v:
                    out[:, i, j] = values(v, n)
    return out


class BatchModels:
    """N model blocks of one topology (arrays with a leading instance axis) -- what
    ``ModelRunner(models=...)`` needs for per-instance matrices.  ``model(i)`` materialises
    instance i as an ordinary ``DiscreteModel``."""

    def __init__(self, data, n, solver=CachingHomotopySolver):
        self.d, self.n, self.solver = data, n, solver

    def __len__(self):
        return self.n

    def model(self, i, solver=None):
        d = self.d
        one = dict(nx=d["nx"], nu=d["nu"], ny=d["ny"], nsub=d["nsub"], nns=d["nns"], nqs=d["nqs"], nps=d["nps"],
                   tables=d["tables"])
        for k in ("a", "b", "c", "x0", "dy", "ey", "fy", "y0"):
            one[k] = d[k][i]
        for k in ("pexps", "dqs", "eqs", "fqprevs", "fqs", "q0s", "init_zs"):
            one[k] = [m[i] for m in d[k]]
        return DiscreteModel(solver=solver or self.solver, _data=one)

    def __getitem__(self, i):
        return self.model(i)

    def __iter__(self):
        return (self.model(i) for i in range(self.n))


def derive_batch(make_circuit, t, component_values, solver=CachingHomotopySolver, decompose_nonlinearity=True,
                 init_on_device=None):
    """``make_circuit(value)`` builds the circuit, calling ``value(name, nominal)`` for every
    component that carries a tolerance (e.g. ``examples.superover(..., value=value)``).
    ``component_values``: dict name -> array[N] of that component's value in each instance
    (components not listed stay nominal).  Returns ``BatchModels``.

    ``init_on_device``: None -> the construction-time solves (``initial_solution`` and the folded
    constant sub-problems) run as a numpy batch on the host; ``True`` or ``dict(lib=..., device=...)``
    -> they run on the GPU, all instances in one batched solve per sub-problem (the dict comes back
    with ``solved`` = number of equations solved there)."""
    dev = None if init_on_device is None else (init_on_device if isinstance(init_on_device, dict) else {})
    names = list(component_values)
    n = len(next(iter(component_values.values()))) if names else 1
    cols = {k: np.asarray(v, dtype=np.float64) for k, v in component_values.items()}
    seen = []

    def value(name, nominal):
        if name not in cols:
            return nominal
        # structure instance: generic (pairwise unrelated) values close to the nominal ones, so
        # that nothing cancels by coincidence (the nominal circuit may hold matched components)
        seen.append(name)
        k = len(seen)
        generic = Fraction(nominal) * (1 + Fraction(2 * k + 1, 9973))
        return BVal(generic, DD(np.concatenate(([float(generic)], cols[name]))))

    circ = make_circuit(value)
    missing = set(names) - set(seen)
    if missing:
        raise KeyError(f"components never requested by make_circuit: {sorted(missing)}")
    nb = n + 1                                  # index 0 = the structure instance itself
    t = Fraction(t) if not isinstance(t, Fraction) else t

    mats = D.model_matrices(circ, t)
    elems = list(circ.elements.values())
    nns = [e.nn for e in elems]
    nqs = [e.nq for e in elems]
    nl_elems = D.nldecompose(mats, nns, nqs) if decompose_nonlinearity else [[i for i, k in enumerate(nns) if k > 0]]
    model_nns = [sum(nns[e] for e in nles) for nles in nl_elems]
    qr = D.consecranges(nqs)
    model_qidxs = [[r for e in nles for r in qr[e]] for nles in nl_elems]
    mats.update(D.split_nl_model_matrices(mats, model_qidxs, model_nns))
    mats = D.reduce_pdims(mats)
    tables = [circ.nonlinear_table(nles) for nles in nl_elems]

    def init_all():
        zs = [np.zeros((nb, k)) for k in model_nns]
        for idx in range(len(tables)):
            if not model_nns[idx]:
                continue
            nq = len(mats["q0s"][idx])
            zall = np.concatenate(zs, axis=1)
            fqprev = _arr(mats["fqprev_fulls"][idx], nb, (nq, zall.shape[1]))
            q = _arr(mats["q0s"][idx], nb, (nq,)) + np.einsum("nqj,nj->nq", fqprev, zall)
            fq = _arr(mats["fqs"][idx], nb, (nq, model_nns[idx]))
            zs[idx] = _initial_solution_batch(tables[idx], fq, q, model_nns[idx], device=dev)
        return zs

    init_zs = init_all()
    while True:                                  # constant sub-problems (src/ACME.jl:196-228)
        const_idxs = [i for i, d in enumerate(mats["dqs"]) if len(d) == 0]
        if not const_idxs:
            break
        zr = D.consecranges(model_nns)
        const_z = [c for i in const_idxs for c in zr[i]]
        varying_z = [c for c in range(sum(model_nns)) if c not in const_z]
        zc_arr = np.concatenate([init_zs[i] for i in const_idxs], axis=1)       # [nb, len(const_z)]
        # the folded constants enter as BVal: structure value = the structure instance's solution
        zc = [BVal(Fraction(float(zc_arr[0, j])), DD(zc_arr[:, j].copy())) for j in range(zc_arr.shape[1])]
        for idx in range(len(mats["q0s"])):
            fp = mats["fqprev_fulls"][idx]
            add = rm.matvec(rm.cols(fp, const_z), zc)
            mats["q0s"][idx] = [a + b for a, b in zip(mats["q0s"][idx], add)]
            mats["fqprev_fulls"][idx] = rm.cols(fp, varying_z)
        mats["x0"] = [a + b for a, b in zip(mats["x0"], rm.matvec(rm.cols(mats["c"], const_z), zc))]
        mats["y0"] = [a + b for a, b in zip(mats["y0"], rm.matvec(rm.cols(mats["fy"], const_z), zc))]
        for key in ("q0s", "dq_fulls", "eq_fulls", "fqs", "fqprev_fulls"):
            mats[key] = [m for i, m in enumerate(mats[key]) if i not in const_idxs]
        init_zs = [m for i, m in enumerate(init_zs) if i not in const_idxs]
        model_nns = [m for i, m in enumerate(model_nns) if i not in const_idxs]
        tables = [m for i, m in enumerate(tables) if i not in const_idxs]
        nl_elems = [m for i, m in enumerate(nl_elems) if i not in const_idxs]
        mats["fy"] = rm.cols(mats["fy"], varying_z)
        mats["c"] = rm.cols(mats["c"], varying_z)
        mats = D.reduce_pdims(mats)

    nx, nu, ny = mats["nx"], mats["nu"], mats["ny"]
    nnt = sum(model_nns)
    nqs_ = [len(m) for m in mats["pexps"]]
    nps_ = [len(m) for m in mats["dqs"]]
    s = slice(1, None)                           # drop the structure instance
    data = dict(
        nx=nx, nu=nu, ny=ny, nsub=len(tables), nns=list(model_nns), nqs=nqs_, nps=nps_, tables=tables,
        a=_arr(mats["a"], nb, (nx, nx))[s], b=_arr(mats["b"], nb, (nx, nu))[s], c=_arr(mats["c"], nb, (nx, nnt))[s],
        x0=_arr(mats["x0"], nb, (nx,))[s],
        dy=_arr(mats["dy"], nb, (ny, nx))[s], ey=_arr(mats["ey"], nb, (ny, nu))[s], fy=_arr(mats["fy"], nb, (ny, nnt))[s],
        y0=_arr(mats["y0"], nb, (ny,))[s],
        pexps=[_arr(m, nb, (nqs_[k], nps_[k]))[s] for k, m in enumerate(mats["pexps"])],
        dqs=[_arr(m, nb, (nps_[k], nx))[s] for k, m in enumerate(mats["dqs"])],
        eqs=[_arr(m, nb, (nps_[k], nu))[s] for k, m in enumerate(mats["eqs"])],
        fqprevs=[_arr(m, nb, (nps_[k], nnt))[s] for k, m in enumerate(mats["fqprevs"])],
        fqs=[_arr(m, nb, (nqs_[k], model_nns[k]))[s] for k, m in enumerate(mats["fqs"])],
        q0s=[_arr(m, nb, (nqs_[k],))[s] for k, m in enumerate(mats["q0s"])],
        init_zs=[z[s] for z in init_zs],
    )
    return BatchModels(data, n, solver)
