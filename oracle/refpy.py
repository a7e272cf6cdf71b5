"""ctypes binding of the CPU oracle (oracle/libacme_ref.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product package acme_jl_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}      # resolved path -> loaded library (ACME_REF_LIB may change between calls: bench.py's native leg)

SOLVER_IDS = {"SimpleSolver": 0, "HomotopySolver{SimpleSolver}": 1,
              "HomotopySolver{CachingSolver{SimpleSolver}}": 2}


class Report(C.Structure):
    _fields_ = [("n_warn", C.c_longlong), ("first_nonconverged", C.c_longlong),
                ("first_nonfinite", C.c_longlong), ("iters_total", C.c_longlong),
                ("iters_max", C.c_longlong), ("lu_swaps", C.c_longlong),
                ("lu_count", C.c_longlong)]


def build():
    """(Re)build libacme_ref.so from acme_ref.c with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    # ACME_REF_LIB: file name of the build to load (bench.py times a -O3 -march=native build of the
    # same source as a second CPU leg); looked up on EVERY call and cached per resolved path, so that a
    # process which already holds the default build still gets the one it asks for
    path = os.path.realpath(os.path.join(_HERE, os.environ.get("ACME_REF_LIB", "libacme_ref.so")))
    if path not in _LIBS:
        src = os.path.join(_HERE, "acme_ref.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.acme_ref_model_create.restype = vp
        L.acme_ref_model_create.argtypes = [C.c_int] * 4 + [dp] * 8
        L.acme_ref_model_add_subproblem.restype = C.c_int
        L.acme_ref_model_add_subproblem.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [dp] * 7 + \
            [C.c_int, ip, ip, ip, dp]
        L.acme_ref_model_destroy.argtypes = [vp]
        L.acme_ref_runner_create.restype = vp
        L.acme_ref_runner_create.argtypes = [vp, C.c_int]
        L.acme_ref_runner_destroy.argtypes = [vp]
        L.acme_ref_set_resabstol.argtypes = [vp, C.c_double]
        L.acme_ref_set_maxiter.argtypes = [vp, C.c_int]
        L.acme_ref_set_cache_limit.argtypes = [vp, C.c_int]
        L.acme_ref_run.restype = C.c_int
        L.acme_ref_run.argtypes = [vp, dp, dp, C.c_longlong, C.POINTER(Report)]
        L.acme_ref_get_x.argtypes = [vp, dp]
        L.acme_ref_set_x.argtypes = [vp, dp]
        L.acme_ref_get_origin.argtypes = [vp, C.c_int, dp, dp]
        L.acme_ref_set_origin.argtypes = [vp, C.c_int, dp, dp]
        L.acme_ref_solve.argtypes = [vp, C.c_int, dp, dp, ip, ip]
        L.acme_ref_lu_factor.restype = C.c_int
        L.acme_ref_lu_factor.argtypes = [C.c_int, dp, ip]
        L.acme_ref_lu_solve.argtypes = [C.c_int, dp, ip, dp]
        L.acme_ref_eval_element.argtypes = [C.c_int, dp, dp, dp, dp]
        L.acme_path = path
        _LIBS[path] = L
    return _LIBS[path]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _fa(a):
    return np.asfortranarray(a, dtype=np.float64)


class RefModel:
    """Oracle-side copy of a DiscreteModel (duck-typed on acme_jl_amd.model.DiscreteModel)."""

    def __init__(self, model):
        L = lib()
        self.nx, self.nu, self.ny, self.nn_total = model.nx, model.nu, model.ny, model.nn_total
        self.subs = [(s.nn, s.nq, s.np) for s in model.subs]
        keep = [_fa(model.a), _fa(model.b), _fa(model.c), _fa(model.x0), _fa(model.dy),
                _fa(model.ey), _fa(model.fy), _fa(model.y0)]
        self.h = L.acme_ref_model_create(model.nx, model.nu, model.ny, model.nn_total,
                                         *[_dp(k) for k in keep])
        for s in model.subs:
            kind, qoff, roff, par = s.elem_arrays()
            mats = [_fa(s.pexp), _fa(s.dq), _fa(s.eq), _fa(s.fqprev), _fa(s.fq), _fa(s.q0),
                    _fa(s.init_z)]
            par = np.ascontiguousarray(par)
            L.acme_ref_model_add_subproblem(self.h, s.nn, s.nq, s.np, *[_dp(k) for k in mats],
                                            len(s.table), _ip(kind), _ip(qoff), _ip(roff),
                                            _dp(par))

    def __del__(self):
        if getattr(self, "h", None):
            lib().acme_ref_model_destroy(self.h)
            self.h = None


class RefRunner:
    """ModelRunner on the oracle: ``run(u) -> y`` == ``run!(runner, u)``."""

    def __init__(self, model, solver=None):
        self.model = model if isinstance(model, RefModel) else RefModel(model)
        if solver is None:
            solver = getattr(model, "solver", "HomotopySolver{SimpleSolver}")
        sid = SOLVER_IDS[solver] if isinstance(solver, str) else int(solver)
        self.h = lib().acme_ref_runner_create(self.model.h, sid)
        if not self.h:
            raise ValueError("unsupported solver")
        self.report = Report()

    def __del__(self):
        if getattr(self, "h", None):
            lib().acme_ref_runner_destroy(self.h)
            self.h = None

    def set_resabstol(self, tol):
        lib().acme_ref_set_resabstol(self.h, float(tol))

    def set_maxiter(self, n):
        lib().acme_ref_set_maxiter(self.h, int(n))

    def set_cache_limit(self, n):
        """CachingSolver with a FIFO store bounded to ``n`` solutions (0: unbounded, the reference)."""
        lib().acme_ref_set_cache_limit(self.h, int(n))

    def run(self, u, raise_on_nonfinite=True):
        m = self.model
        u = np.asfortranarray(u, dtype=np.float64)
        if u.ndim != 2 or u.shape[0] != m.nu:
            raise ValueError(f"input matrix has {u.shape[0]} rows, but model has {m.nu} inputs")
        T = u.shape[1]
        y = np.full((m.ny, T), np.nan, order="F")
        rc = lib().acme_ref_run(self.h, _dp(u), _dp(y), T, C.byref(self.report))
        if rc and raise_on_nonfinite:
            raise RuntimeError("Failed to converge while solving non-linear equation, "
                               "got non-finite result.")
        return y

    @property
    def x(self):
        x = np.zeros(self.model.nx)
        lib().acme_ref_get_x(self.h, _dp(x))
        return x

    @x.setter
    def x(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        lib().acme_ref_set_x(self.h, _dp(v))

    def get_origin(self, sub=0):
        nn, nq, np_ = self.model.subs[sub]
        p, z = np.zeros(np_), np.zeros(nn)
        lib().acme_ref_get_origin(self.h, sub, _dp(p), _dp(z))
        return p, z

    def set_origin(self, sub, p, z):
        p = np.ascontiguousarray(p, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        lib().acme_ref_set_origin(self.h, sub, _dp(p), _dp(z))

    def solve(self, p, sub=0):
        nn, nq, np_ = self.model.subs[sub]
        p = np.ascontiguousarray(p, dtype=np.float64)
        z = np.zeros(nn)
        conv, iters = C.c_int(0), C.c_int(0)
        lib().acme_ref_solve(self.h, sub, _dp(p), _dp(z), C.byref(conv), C.byref(iters))
        return z, bool(conv.value), iters.value


def lu_factor(A):
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    ipiv = np.zeros(n, dtype=np.int32)
    ok = lib().acme_ref_lu_factor(n, _dp(A), _ip(ipiv))
    return bool(ok), A, ipiv


def lu_solve(f, ipiv, b):
    x = np.array(b, dtype=np.float64)
    f = np.asfortranarray(f)
    lib().acme_ref_lu_solve(f.shape[0], _dp(f), _ip(ipiv), _dp(x))
    return x
