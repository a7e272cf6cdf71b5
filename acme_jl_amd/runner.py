"""ModelRunner / run: the host-side mirror of the reference's runner API over the C ABI.

Reference interface mirrored (src/ACME.jl):
  * ``ModelRunner(model, showprogress)`` ............ :570-604
  * ``run!(runner, u)`` / ``run!(runner, y, u)`` ..... :619-664 (+ ``checkiosizes`` :625-635)
  * ``run!(model, u)`` ............................... :567-568
  * failure policy of ``step!`` ....................... :688-694
  * ``set_resabstol!`` / extrapolation origin ......... src/solvers.jl:181-198

The difference to the reference is the batch axis: one runner advances ``n_instances``
independent copies of the circuit in lock step on one GPU.  All compute happens in
``libacme_hip.so`` (hand-written HIP for gfx950).  There is NO CPU fallback: if the
library or a GPU is missing, constructing a runner raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np

from .model import SOLVER_IDS, CachingHomotopySolver, DiscreteModel

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACME_HIP_LIB: developer override used to A/B kernel build variants (tools/variants.sh)
DEFAULT_LIBRARY = os.environ.get("ACME_HIP_LIB") or os.path.join(_HERE, "csrc", "libacme_hip.so")

ACME_MEM_HOST, ACME_MEM_DEVICE = 0, 1


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (src/ACME.jl:625-635)."""


class AcmeError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("solver", C.c_int), ("tol", C.c_double), ("maxiter", C.c_int),
                ("device", C.c_int), ("per_instance_matrices", C.c_int)]


class Report(C.Structure):
    _fields_ = [("n_warn", C.c_longlong), ("first_nonconverged", C.c_longlong),
                ("first_nonfinite", C.c_longlong), ("iters_total", C.c_longlong),
                ("iters_max", C.c_longlong)]


# every symbol include/acme_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "acme_last_error", "acme_device_count", "acme_default_options", "acme_model_create",
    "acme_model_add_subproblem", "acme_model_set_row_order", "acme_model_destroy",
    "acme_model_kernel_shape", "acme_model_kernel_variant",
    "acme_batch_create", "acme_batch_destroy", "acme_batch_kernel_variant", "acme_batch_set_matrices", "acme_batch_run", "acme_batch_run_const",
    "acme_batch_run_async", "acme_batch_wait", "acme_batch_set_host_retention", "acme_batch_release_host_buffers", "acme_batch_set_progress_callback", "acme_batch_set_isolation",
    "acme_batch_set_balance", "acme_batch_get_placement",
    "acme_batch_solve", "acme_batch_get_extrapolation_jacobian", "acme_batch_last_kernel_ms", "acme_batch_kernel_time", "acme_batch_get_report", "acme_batch_reset_report",
    "acme_batch_set_resabstol", "acme_batch_get_state", "acme_batch_set_state",
]


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  If libacme_hip.so pulled in the
    system ROCm runtime first, torch would later be bound to that second, mismatching runtime
    ("No HIP GPUs are available").  Loading torch's copy first -- without importing torch --
    makes both share one HIP runtime regardless of import order.  Without torch installed
    (e.g. the Julia ccall route) the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            return None
    return None


# acme_progress_fn(user, samples_done, samples_total)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_longlong, C.c_longlong)


class Library:
    """ctypes binding of one shared library implementing include/acme_hip.h."""

    def __init__(self, path=DEFAULT_LIBRARY):
        if os.path.abspath(path) == os.path.abspath(DEFAULT_LIBRARY):
            self._hip_rt = _preload_torch_hip_runtime()
        if not os.path.exists(path):
            raise AcmeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                " (hipcc --offload-arch=gfx950).  acme_jl_amd has no CPU fallback.")
        self.path = path
        L = self.L = C.CDLL(path)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.acme_last_error.restype = C.c_char_p
        L.acme_device_count.restype = C.c_int
        L.acme_default_options.argtypes = [C.POINTER(Options)]
        L.acme_model_create.argtypes = [C.c_int] * 4 + [dp] * 8 + [C.POINTER(vp)]
        L.acme_model_add_subproblem.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [dp] * 7 + \
            [C.c_int, ip, ip, ip, dp]
        L.acme_model_set_row_order.argtypes = [vp, C.c_int, ip, C.c_int]
        L.acme_model_destroy.argtypes = [vp]
        L.acme_model_destroy.restype = None
        L.acme_model_kernel_shape.argtypes = [vp, ip]
        L.acme_model_kernel_variant.argtypes = [vp, ip, ip]
        L.acme_batch_create.argtypes = [vp, C.c_longlong, C.POINTER(Options), C.POINTER(vp)]
        L.acme_batch_destroy.argtypes = [vp]
        L.acme_batch_destroy.restype = None
        L.acme_batch_kernel_variant.argtypes = [vp, ip, ip]
        L.acme_batch_set_matrices.argtypes = [vp, C.c_longlong, C.c_longlong, C.POINTER(vp)]
        L.acme_batch_run.argtypes = [vp, vp, vp, C.c_longlong, C.c_int, vp]
        L.acme_batch_run_const.argtypes = [vp, vp, vp, C.c_ulonglong, vp, C.c_longlong, C.c_int, vp]
        L.acme_batch_run_async.argtypes = [vp, vp, vp, C.c_longlong, C.c_int, vp]
        L.acme_batch_wait.argtypes = [vp]
        L.acme_batch_set_host_retention.argtypes = [vp, C.c_int]
        L.acme_batch_release_host_buffers.argtypes = [vp]
        L.acme_batch_set_isolation.argtypes = [vp, C.c_double]
        L.acme_batch_set_balance.argtypes = [vp, C.c_int]
        L.acme_batch_get_placement.argtypes = [vp, ip, C.POINTER(C.c_longlong)]
        L.acme_batch_set_progress_callback.argtypes = [vp, PROGRESS_FN, vp]
        L.acme_batch_solve.argtypes = [vp, C.c_int, dp, dp, ip, ip, C.c_int, vp]
        L.acme_batch_get_extrapolation_jacobian.argtypes = [vp, C.c_int, dp, C.c_int, vp]
        L.acme_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.acme_batch_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
        L.acme_batch_get_report.argtypes = [vp, C.POINTER(Report)]
        L.acme_batch_reset_report.argtypes = [vp]
        L.acme_batch_set_resabstol.argtypes = [vp, C.c_double]
        L.acme_batch_get_state.argtypes = [vp, dp, dp, dp]
        L.acme_batch_set_state.argtypes = [vp, dp, dp, dp]

    def check(self, rc):
        if rc < 0:
            raise AcmeError(self.L.acme_last_error().decode())
        return rc

    def device_count(self):
        return self.L.acme_device_count()


_DEFAULT = None


def default_library():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Library(DEFAULT_LIBRARY)
    return _DEFAULT


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _fa(a):
    return np.asfortranarray(a, dtype=np.float64)


class _ModelHandle:
    """acme_model* built from a DiscreteModel."""

    def __init__(self, lib, model):
        self.lib = lib
        h = C.c_void_p()
        keep = [_fa(model.a), _fa(model.b), _fa(model.c), _fa(model.x0), _fa(model.dy),
                _fa(model.ey), _fa(model.fy), _fa(model.y0)]
        lib.check(lib.L.acme_model_create(model.nx, model.nu, model.ny, model.nn_total,
                                          *[_dp(k) for k in keep], C.byref(h)))
        self.h = h
        for k, s in enumerate(model.subs):
            kind, qoff, roff, par = s.elem_arrays()
            mats = [_fa(s.pexp), _fa(s.dq), _fa(s.eq), _fa(s.fqprev), _fa(s.fq), _fa(s.q0),
                    _fa(s.init_z)]
            par = np.ascontiguousarray(par)
            lib.check(lib.L.acme_model_add_subproblem(
                h, s.nn, s.nq, s.np, *[_dp(k) for k in mats], len(s.table), _ip(kind),
                _ip(qoff), _ip(roff), _dp(par)))
            if s.row_order is not None:
                ro = np.ascontiguousarray(s.row_order, dtype=np.int32)
                lib.check(lib.L.acme_model_set_row_order(h, k, _ip(ro), len(ro)))

    def kernel_shape(self):
        dims = (C.c_int * 6)()
        self.lib.check(self.lib.L.acme_model_kernel_shape(self.h, dims))
        return tuple(dims)

    def kernel_variant(self):
        """(condensed_rows, generic): rows eliminated ahead of the Newton iteration, run-time-sized kernel"""
        nl, gen = C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.L.acme_model_kernel_variant(self.h, C.byref(nl), C.byref(gen)))
        return nl.value, bool(gen.value)

    def kernel_family(self):
        """"tuned" (an instantiated shape), "coop" (run-time-sized, one instance per 16 lanes, working arrays in LDS:
        csrc/acme_coop.h) or "generic" (run-time-sized, one lane per instance: csrc/acme_generic.h)"""
        gen = C.c_int(0)
        self.lib.check(self.lib.L.acme_model_kernel_variant(self.h, None, C.byref(gen)))
        return ("tuned", "generic", "coop")[gen.value]

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.L.acme_model_destroy(self.h)
            self.h = None


def _print_progress(done, total):
    import sys
    sys.stderr.write("\rrun!: %3d %% (%d / %d samples)%s" % (100 * done // max(total, 1), done, total, "\n" if done >= total else ""))
    sys.stderr.flush()


class ModelRunner:
    """``ModelRunner(model, showprogress)`` for ``n_instances`` parallel copies of a model.

    ``models``: optional list of ``n_instances`` DiscreteModels with the same circuit
    topology but different component values (per-instance matrices, e.g. Monte-Carlo
    component tolerances); the element tables must be identical.
    """

    def __init__(self, model, n_instances=1, showprogress=False, device=None, models=None,
                 lib=None):
        if not isinstance(model, DiscreteModel):
            raise TypeError("model must be a DiscreteModel")
        self.lib = lib or default_library()
        if self.lib.device_count() <= 0:
            raise AcmeError("no HIP device available; acme_jl_amd has no CPU fallback")
        self.model = model
        self.n = int(n_instances)
        # ``showprogress``: True prints a progress line per time slice of a host-buffer run (the reference wraps
        # its sample loop in @showprogress, src/ACME.jl:587-604,653); a callable gets (samples_done, samples_total)
        self.showprogress = showprogress
        self._mh = _ModelHandle(self.lib, model)
        o = Options()
        self.lib.L.acme_default_options(C.byref(o))
        o.solver = SOLVER_IDS[model.solver]
        o.device = -1 if device is None else int(device)
        o.per_instance_matrices = 1 if models is not None else 0
        h = C.c_void_p()
        self.lib.check(self.lib.L.acme_batch_create(self._mh.h, self.n, C.byref(o), C.byref(h)))
        self.h = h
        self._warned = 0
        self._progress_cb = None
        if showprogress:
            fn = showprogress if callable(showprogress) else _print_progress
            self._progress_cb = PROGRESS_FN(lambda user, done, total: fn(int(done), int(total)))    # (kept alive here)
            self.lib.check(self.lib.L.acme_batch_set_progress_callback(self.h, self._progress_cb, None))
        if models is not None:
            self.set_models(0, models)

    def set_isolation(self, iters_per_sample):
        """Run the instances that needed more than ``iters_per_sample`` Newton iterations per sample over the previous
        run in a launch of their own (``acme_batch_set_isolation``): device-pointer runs then complete on the caller's
        stream for the others, ``wait()`` completes the slow ones.  0 switches it off."""
        self.lib.check(self.lib.L.acme_batch_set_isolation(self.h, float(iters_per_sample)))

    def set_balance(self, mode=-1):
        """Placement of the waves by their measured cost (``acme_batch_set_balance``): -1 the library decides
        (default), 0 off, 1 on.  Results do not depend on it (bit-identical)."""
        self.lib.check(self.lib.L.acme_batch_set_balance(self.h, int(mode)))
        return self

    def placement(self):
        """slot -> instance of the last launch (``acme_batch_get_placement``): -1 = empty slot; the identity while
        nothing is placed"""
        n = C.c_longlong(0)
        self.lib.check(self.lib.L.acme_batch_get_placement(self.h, None, C.byref(n)))
        out = np.empty(n.value, dtype=np.int32)
        self.lib.check(self.lib.L.acme_batch_get_placement(self.h, _ip(out), None))
        return out

    def set_host_retention(self, keep=True):
        """``acme_batch_set_host_retention``: promise that the host arrays handed to ``run_async`` / ``run(layout="abi")``
        stay allocated until others are passed, ``release_host_buffers()`` is called or the runner goes -- they are then
        page-locked once and runs are streamed at the device-resident rate.  Off by default: ``run`` works on per-call
        temporaries, which must never stay registered."""
        self.lib.check(self.lib.L.acme_batch_set_host_retention(self.h, 1 if keep else 0))
        self._retain = bool(keep)
        if not keep:
            self._held = None

    def release_host_buffers(self):
        """Un-page-lock the arrays of the last host-buffer run (``acme_batch_release_host_buffers``)."""
        self.lib.check(self.lib.L.acme_batch_release_host_buffers(self.h))
        self._held = None

    def _hold(self, *arrays):
        # a retaining runner keeps the arrays it handed to the library alive for as long as they may be page-locked:
        # until the next call's arrays replace them, release_host_buffers() or the batch's destruction
        if getattr(self, "_retain", False):
            self._held = arrays

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.L.acme_batch_destroy(self.h)       # (un-page-locks whatever is held: before the arrays go)
            self.h = None
        self._held = None

    # ---- per-instance matrices ----------------------------------------------------------
    def set_models(self, first, models, chunk=1024):
        """Give instances first.. their own model blocks (``acme_batch_set_matrices``); ``models``
        is any iterable of DiscreteModels, e.g. ``montecarlo.BatchModels``."""
        hs = []

        def flush():
            nonlocal first, hs
            if hs:
                arr = (C.c_void_p * len(hs))(*[m.h for m in hs])
                self.lib.check(self.lib.L.acme_batch_set_matrices(self.h, first, len(hs), arr))
                first += len(hs)
                hs = []
        for m in models:
            hs.append(_ModelHandle(self.lib, m))
            if len(hs) == chunk:
                flush()
        flush()

    # ---- run! ---------------------------------------------------------------------------
    def _check_io(self, u_rows, y_rows, ucols, ycols):
        m = self.model
        if u_rows != m.nu:
            raise DimensionMismatch(f"input matrix has {u_rows} rows, but model has {m.nu} inputs")
        if y_rows != m.ny:
            raise DimensionMismatch(f"output matrix has {y_rows} rows, but model has {m.ny} outputs")
        if ucols != ycols:
            raise DimensionMismatch(
                f"input matrix has {ucols} columns, output matrix has {ycols} columns")

    def run(self, u, y=None, check=True, time_major=False):
        """``run!(runner, u)`` / ``run!(runner, y, u)`` with numpy arrays.

        ``u``: (nu, T) for a single instance, else (N, nu, T); returns ``y`` of shape
        (ny, T) resp. (N, ny, T).  Raises like the reference when an instance hits a
        non-finite result and warns on convergence failures (unless ``check=False``).

        ``time_major=True``: ``u`` is (N, T, nu) and ``y`` (N, T, ny) -- the memory layout of the
        C ABI (and of Julia's nu x T matrices), passed through without the two transposing copies
        the default shapes cost when nu or ny exceed 1."""
        m = self.model
        u = np.asarray(u, dtype=np.float64)
        if time_major:
            if u.ndim != 3 or u.shape[0] != self.n:
                raise DimensionMismatch(f"input must have shape ({self.n}, T, {m.nu})")
            T = u.shape[1]
            if y is not None:
                if not (isinstance(y, np.ndarray) and y.dtype == np.float64 and y.flags.c_contiguous):
                    raise TypeError("y must be a C-contiguous float64 array")
                if y.ndim != 3 or y.shape[0] != self.n:
                    raise DimensionMismatch(f"output must have shape ({self.n}, T, {m.ny})")
                self._check_io(u.shape[2], y.shape[2], T, y.shape[1])
            else:
                self._check_io(u.shape[2], m.ny, T, T)
                y = np.empty((self.n, T, m.ny), dtype=np.float64)
            ub = np.ascontiguousarray(u)
            self.lib.check(self.lib.L.acme_batch_run(self.h, ub.ctypes.data, y.ctypes.data, T, ACME_MEM_HOST, None))
            self._hold(ub, y)
            if check:
                self.check()
            return y
        single = u.ndim == 2
        if single:
            if self.n != 1:
                raise DimensionMismatch("2-D input given to a runner with more than one instance")
            u = u[None]
        if u.ndim != 3 or u.shape[0] != self.n:
            raise DimensionMismatch(f"input must have shape ({self.n}, {m.nu}, T)")
        T = u.shape[2]
        if y is not None:
            y = np.asarray(y)
            yy = y[None] if single else y
            self._check_io(u.shape[1], yy.shape[1], T, yy.shape[2])
        else:
            self._check_io(u.shape[1], m.ny, T, T)
        ub = np.ascontiguousarray(np.transpose(u, (0, 2, 1)))        # [N][T][nu]
        yb = np.empty((self.n, T, m.ny), dtype=np.float64)
        self.lib.check(self.lib.L.acme_batch_run(
            self.h, ub.ctypes.data, yb.ctypes.data, T, ACME_MEM_HOST, None))
        self._hold(ub, yb)
        out = np.transpose(yb, (0, 2, 1))
        if y is not None:
            (y[None] if single else y)[...] = out
        if check:
            self.check()
        if y is not None:
            return y
        return np.asfortranarray(out[0]) if single else np.ascontiguousarray(out)

    def run_const(self, u_var, u_const, const_rows, y=None, check=True):
        """``run!`` with constant input rows (``acme_batch_run_const``): ``const_rows`` names the input rows that keep one
        value per instance for the whole call -- ``u_const`` (N, nu): their values (the other entries are ignored) --,
        ``u_var`` (N, T, nu_var) holds the remaining rows in row order (the ABI's time-major layout).  A sweep over
        potentiometer positions then moves a quarter of the bytes over the bus; the results are those of ``run`` on the
        materialised input, bit for bit.  Returns y (N, T, ny)."""
        m = self.model
        rows = sorted(set(int(k) for k in const_rows))
        if any(k < 0 or k >= m.nu for k in rows):
            raise DimensionMismatch(f"constant rows {rows} of a model with {m.nu} inputs")
        mask = 0
        for k in rows:
            mask |= 1 << k
        nuv = m.nu - len(rows)
        u_var = np.ascontiguousarray(u_var, dtype=np.float64)
        u_const = np.ascontiguousarray(u_const, dtype=np.float64)
        if u_var.ndim != 3 or u_var.shape[0] != self.n or u_var.shape[2] != nuv:
            raise DimensionMismatch(f"u_var must have shape ({self.n}, T, {nuv})")
        if u_const.shape != (self.n, m.nu):
            raise DimensionMismatch(f"u_const must have shape ({self.n}, {m.nu})")
        T = u_var.shape[1]
        if y is None:
            y = np.empty((self.n, T, m.ny), dtype=np.float64)
        elif not (isinstance(y, np.ndarray) and y.dtype == np.float64 and y.flags.c_contiguous and y.shape == (self.n, T, m.ny)):
            raise DimensionMismatch(f"y must be a C-contiguous float64 array of shape ({self.n}, {T}, {m.ny})")
        self.lib.check(self.lib.L.acme_batch_run_const(self.h, u_var.ctypes.data, u_const.ctypes.data, mask, y.ctypes.data, T, ACME_MEM_HOST, None))
        self._hold(u_var, u_const, y)
        if check:
            self.check()
        return y

    def run_async(self, u, y):
        """``acme_batch_run_async`` on host buffers in the ABI's layout: ``u`` (N, T, nu) and ``y``
        (N, T, ny), C-contiguous float64 (slices of larger arrays along the first axis are fine).
        Returns at once; ``wait()`` joins the run.  The caller keeps ``u`` / ``y`` alive until then."""
        m = self.model
        for a, cols, what in ((u, m.nu, "u"), (y, m.ny, "y")):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
                raise TypeError(f"{what} must be a C-contiguous float64 array")
            if a.ndim != 3 or a.shape[0] != self.n or a.shape[2] != cols:
                raise DimensionMismatch(f"{what} must have shape ({self.n}, T, {cols})")
        self._check_io(u.shape[2], y.shape[2], u.shape[1], y.shape[1])
        self._inflight = (u, y)
        self._hold(u, y)
        self.lib.check(self.lib.L.acme_batch_run_async(self.h, u.ctypes.data, y.ctypes.data, u.shape[1],
                                                       ACME_MEM_HOST, None))

    def wait(self, check=True):
        """``acme_batch_wait``: join the run started by ``run_async``; raises what it failed with."""
        try:
            self.lib.check(self.lib.L.acme_batch_wait(self.h))
        finally:
            self._inflight = None
        if check:
            self.check()

    def run_device(self, u_ptr, y_ptr, T, stream=None):
        """Raw asynchronous launch: ``u_ptr``/``y_ptr`` are device addresses of
        [N][T][nu] / [N][T][ny] float64 buffers on this runner's GPU."""
        self.lib.check(self.lib.L.acme_batch_run(
            self.h, C.c_void_p(u_ptr), C.c_void_p(y_ptr), int(T), ACME_MEM_DEVICE,
            C.c_void_p(stream or 0)))

    def run_torch(self, u, y=None):
        """``u``: torch float64 CUDA tensor (N, T, nu); returns/fills ``y`` (N, T, ny).
        Launches on torch's current stream, does not synchronise."""
        import torch
        m = self.model
        if u.dtype != torch.float64 or not u.is_cuda or not u.is_contiguous():
            raise TypeError("u must be a contiguous float64 CUDA tensor")
        if u.dim() != 3 or u.shape[0] != self.n or u.shape[2] != m.nu:
            raise DimensionMismatch(f"u must have shape ({self.n}, T, {m.nu})")
        T = u.shape[1]
        if y is None:
            y = torch.empty((self.n, T, m.ny), dtype=torch.float64, device=u.device)
        elif tuple(y.shape) != (self.n, T, m.ny) or y.dtype != torch.float64 or not y.is_contiguous():
            raise DimensionMismatch(f"y must be a contiguous float64 tensor of shape ({self.n}, {T}, {m.ny})")
        stream = torch.cuda.current_stream(u.device).cuda_stream
        self.run_device(u.data_ptr(), y.data_ptr(), T, stream)
        return y

    # ---- reports, failure policy ----------------------------------------------------------
    def reports(self):
        r = (Report * self.n)()
        self.lib.check(self.lib.L.acme_batch_get_report(self.h, r))
        return r

    def report_arrays(self):
        r = self.reports()
        a = np.frombuffer(r, dtype=np.int64).reshape(self.n, 5).copy()
        return dict(n_warn=a[:, 0], first_nonconverged=a[:, 1], first_nonfinite=a[:, 2],
                    iters_total=a[:, 3], iters_max=a[:, 4])

    def check(self):
        """Apply step!'s policy (src/ACME.jl:688-694) to the accumulated reports."""
        ra = self.report_arrays()
        if (ra["first_nonfinite"] >= 0).any():
            i = int(np.argmax(ra["first_nonfinite"] >= 0))
            raise AcmeError("Failed to converge while solving non-linear equation, got non-finite "
                            f"result. (instance {i}, sample {int(ra['first_nonfinite'][i])})")
        nw = int(ra["n_warn"].sum())
        if nw > self._warned:
            warnings.warn("Failed to converge while solving non-linear equation.")
            self._warned = nw

    def reset_report(self):
        self.lib.check(self.lib.L.acme_batch_reset_report(self.h))
        self._warned = 0

    def last_kernel_ms(self):
        ms = C.c_float()
        self.lib.check(self.lib.L.acme_batch_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def kernel_time(self, reset=False):
        """(total ms, launches) of the kernels since the last reset, from HIP events."""
        ms, n = C.c_double(), C.c_longlong()
        self.lib.check(self.lib.L.acme_batch_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # ---- solver plugin surface ----------------------------------------------------------------
    def set_resabstol(self, tol):
        """``set_resabstol!`` (src/solvers.jl:181,262)."""
        self.lib.check(self.lib.L.acme_batch_set_resabstol(self.h, float(tol)))

    def solve(self, p, sub=0):
        """Batched ``solve(model.solvers[sub+1], p)`` (src/solvers.jl:207-236, 268-302): ``p`` is
        (N, np); returns ``(z, hasconverged, needediterations)`` with shapes (N, nn), (N,), (N,).
        Uses and updates each instance's extrapolation origin like the reference's solver objects."""
        s = self.model.subs[sub]
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(p, dtype=np.float64), (self.n, s.np)))
        z = np.zeros((self.n, s.nn))
        conv = np.zeros(self.n, dtype=np.int32)
        iters = np.zeros(self.n, dtype=np.int32)
        self.lib.check(self.lib.L.acme_batch_solve(self.h, int(sub), _dp(p), _dp(z), _ip(conv), _ip(iters),
                                                   ACME_MEM_HOST, None))
        return z, conv.astype(bool), iters

    def get_extrapolation_jacobian(self, sub=0):
        """Batched ``get_extrapolation_jacobian(model.solvers[sub+1])`` (src/solvers.jl:198-201):
        (N, nn, np) array of dz/dp = -(J \\ Jp) at every instance's extrapolation origin."""
        s = self.model.subs[sub]
        jac = np.zeros((self.n, s.np, s.nn))          # ABI layout: per instance column-major nn x np
        self.lib.check(self.lib.L.acme_batch_get_extrapolation_jacobian(self.h, int(sub), _dp(jac), ACME_MEM_HOST, None))
        return np.ascontiguousarray(jac.transpose(0, 2, 1))

    def get_state(self):
        """(x, last_p, last_z): model.x and the extrapolation origin of every instance."""
        m = self.model
        x = np.zeros((self.n, m.nx))
        p = np.zeros((self.n, sum(s.np for s in m.subs)))
        z = np.zeros((self.n, sum(s.nn for s in m.subs)))
        self.lib.check(self.lib.L.acme_batch_get_state(self.h, _dp(x), _dp(p), _dp(z)))
        return x, p, z

    def set_state(self, x=None, p=None, z=None):
        def prep(a, cols):
            if a is None:
                return None, None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.n, cols)))
            return a, _dp(a)
        m = self.model
        xa, xp = prep(x, m.nx)
        pa, pp = prep(p, sum(s.np for s in m.subs))
        za, zp = prep(z, sum(s.nn for s in m.subs))
        self.lib.check(self.lib.L.acme_batch_set_state(self.h, xp, pp, zp))

    def kernel_shape(self):
        return self._mh.kernel_shape()

    def kernel_variant(self):
        return self._mh.kernel_variant()

    def kernel_family(self):
        """the kernel family THIS BATCH runs in ("tuned", "generic", "coop"): its model's, unless acme_batch_set_matrices moved
        it (instances with element parameters of their own)"""
        fam = C.c_int(0)
        self.lib.check(self.lib.L.acme_batch_kernel_variant(self.h, None, C.byref(fam)))
        return ("tuned", "generic", "coop")[fam.value]

    def batch_kernel_variant(self):
        """(condensed_rows, family) of the batch as it runs now (``acme_batch_kernel_variant``)"""
        nl, fam = C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.L.acme_batch_kernel_variant(self.h, C.byref(nl), C.byref(fam)))
        return nl.value, ("tuned", "generic", "coop")[fam.value]


class MultiDeviceRunner:
    """N instances of a model spread over several GPUs of ONE node by ONE process: contiguous instance
    ranges (``dist.shard_range``), one ``ModelRunner`` per device, every run started asynchronously on all
    of them (``acme_batch_run_async``) and then joined, each batch reading and writing its own slice of
    the caller's ``u`` / ``y``.  No collective is involved -- the sweep shards perfectly -- so this is the
    multi-GPU path of a host that has no torch.distributed (the Julia binding's ``MultiBatchRunner`` is
    the same thing over the same C ABI).  ``devices``: HIP ordinals, default all visible ones; an ordinal
    may repeat (several batches on one GPU), which is how the path is tested on a single-GPU box."""

    def __init__(self, model, n_instances, devices=None, lib=None, models=None):
        from .dist import shard_range
        self.lib = lib or default_library()
        if devices is None:
            devices = list(range(self.lib.device_count()))
        if not devices:
            raise AcmeError("no HIP device available; acme_jl_amd has no CPU fallback")
        self.model, self.n, self.devices = model, int(n_instances), list(devices)
        self.ranges = [shard_range(self.n, k, len(self.devices)) for k in range(len(self.devices))]
        self.runners = []
        for dev, (lo, hi) in zip(self.devices, self.ranges):
            if hi > lo:
                part = None if models is None else [models[i] for i in range(lo, hi)]
                self.runners.append(ModelRunner(model, hi - lo, device=dev, lib=self.lib, models=part))
            else:
                self.runners.append(None)

    def run(self, u, y=None, check=True):
        """``u``: (N, T, nu) C-contiguous float64 (the ABI's layout); returns / fills ``y`` (N, T, ny)."""
        m = self.model
        u = np.ascontiguousarray(u, dtype=np.float64)
        if u.ndim != 3 or u.shape[0] != self.n or u.shape[2] != m.nu:
            raise DimensionMismatch(f"input must have shape ({self.n}, T, {m.nu})")
        if y is None:
            y = np.empty((self.n, u.shape[1], m.ny), dtype=np.float64)
        elif y.shape != (self.n, u.shape[1], m.ny) or y.dtype != np.float64 or not y.flags.c_contiguous:
            raise DimensionMismatch(f"output must be a C-contiguous float64 array of shape ({self.n}, {u.shape[1]}, {m.ny})")
        started, err = [], None
        for r, (lo, hi) in zip(self.runners, self.ranges):
            if r is not None:
                r.run_async(u[lo:hi], y[lo:hi])
                started.append(r)
        for r in started:                       # join every run, then report the first failure
            try:
                r.wait(check=check)
            except AcmeError as e:
                err = err or e
        if err is not None:
            raise err
        return y

    def report_arrays(self):
        parts = [r.report_arrays() for r in self.runners if r is not None]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def get_state(self):
        parts = [r.get_state() for r in self.runners if r is not None]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


def run(model_or_runner, u, **kw):
    """``run!(model, u)`` / ``run!(runner, u)`` (src/ACME.jl:567-568, 619-623).  The model's
    state persists across calls: the runner is cached on the model object."""
    if isinstance(model_or_runner, ModelRunner):
        return model_or_runner.run(u, **kw)
    model = model_or_runner
    r = getattr(model, "_runner", None)
    if r is None:
        r = model._runner = ModelRunner(model, 1)
    return r.run(u, **kw)
