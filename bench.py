#!/usr/bin/env python
"""bench.py -- throughput of the batched run! hot path on MI355X.

Metric (BASELINE.json): circuit-instance*samples/sec, superover @ 44.1 kHz.
Workload at N GPUs (weak scaling, 8192 instances per GPU): examples/superover.jl with the
three potentiometers as inputs (shared model block, nn=13/nq=29/np=11), one instance per
point of a drive x tone x level grid, 1 kHz unit sine, one "step" = 1 s of audio
(44100 samples) for every instance.  Instances are independent: ranks shard the grid, the
model block is broadcast once over RCCL before the timed region, and the only collective
inside it is the all-reduce of the per-rank solver counters.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     -- algorithmic HBM bytes per launch / kernel time measured with HIP events
  cpu_baseline -- the CPU oracle (a C restatement of the reference algorithm, NOT the Julia
                  reference, which cannot run here) timed on the host cores of this box
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector peak (spec); not in the guide, AMD datasheet
FS = 44100


def grid_inputs(workload, rank, world, n_per_gpu, T):
    """(model fixture name, numpy u-parameters) for this rank's shard of the sweep.

    superover_grid: global grid drive(32*world) x tone(16) x level(16); level varies fastest
    so that the 4 instances sharing a wavefront differ only in the output-stage pot (they
    then need identical Newton iteration counts -> no intra-wave divergence).  drive stops
    short of 1.0: at exactly 1.0 the pot's shorted leg has 0 Ohm, its current is
    indeterminate and the variable-pot model is singular (the reference warns on every
    sample there)."""
    idx = np.arange(n_per_gpu) + rank * n_per_gpu
    total = n_per_gpu * world
    if workload == "superover_grid":
        if total % 256:
            raise SystemExit("superover_grid needs a multiple of 256 instances (16 tone x 16 level per drive value)")
        nd = total // 256
        level = (idx % 16) / 15.0
        tone = ((idx // 16) % 16) / 15.0
        drive = (idx // 256) / float(nd)
        pots = np.stack([drive, tone, level], axis=1)
        return "superover_var", pots, 1.0
    if workload == "diodeclipper_sweep":
        amp = 10.0 ** (-2 + 3 * idx / max(total - 1, 1))
        return "diodeclipper", None, amp
    if workload == "superover_montecarlo":   # BASELINE config 4: models derived in montecarlo_models()
        return "superover_fixed", None, 1.0
    if workload == "birdie_grid":            # BASELINE config 5: amplitude x vol grid, vol fastest
        if total % 128:
            raise SystemExit("birdie_grid needs a multiple of 128 instances (128 vol values per amplitude)")
        na = total // 128
        amp = 10.0 ** (-2 + 2.5 * (idx // 128) / max(na - 1, 1))
        vol = 0.01 + 0.99 * (idx % 128) / 127.0
        return "birdie_var_176k", vol[:, None], amp
    if workload.startswith("clipper_chain_"):       # beyond BASELINE: one sub-problem of that many unknowns (workload_model), amplitude sweep
        return "chain:%d" % (int(workload.rsplit("_", 1)[1]) // 2), None, 10.0 ** (-2 + 2.7 * idx / max(total - 1, 1))
    raise ValueError(workload)


def workload_model(workload, fixture, solver, fs=FS):
    """the workload's DiscreteModel: a committed model-block fixture (tests/golden/*.json: outputs of this repository's
    front end, re-derived by tests/test_frontend.py) or, for clipper_chain_20, derived on the spot (0.2 s).
    clipper_chain_20 is beyond BASELINE: ONE nonlinear sub-problem of 20 unknowns -- the reference's LU is written for
    "sizes up to about 60 x 60" (src/solvers.jl:53-54) and nldecompose! leaves such sub-problems whenever a circuit does not
    decompose -- which runs in the cooperative mid-size kernel (csrc/acme_coop.h); a tenth of a second of audio per step."""
    from acme_jl_amd.model import DiscreteModel
    if fixture and fixture.startswith("chain:"):          # (clipper_chain_20 / _34: ten / seventeen stages)
        from fractions import Fraction
        from acme_jl_amd import examples
        return DiscreteModel(examples.clipper_chain(int(fixture[6:])), Fraction(1, fs), solver, decompose_nonlinearity=False)
    return DiscreteModel.load(os.path.join(ROOT, "tests", "golden", fixture + ".json"), solver=solver)


def montecarlo_models(rank, n_per_gpu, init_on_device=None):
    """BASELINE config 4: fixed-pot superover (1.0, 1.0, 1.0), every resistor, capacitor and pot
    track scaled by 1 + 0.05*U(-1,1), semiconductors nominal.  Every rank derives its own shard
    locally from (seed 20250905, rank) with the structure-replaying front end."""
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.montecarlo import derive_batch
    make = lambda value: examples.superover(1.0, 1.0, 1.0, value=value)     # noqa: E731
    nominal = {}
    make(lambda name, v: nominal.setdefault(name, v))
    rng = np.random.Generator(np.random.PCG64([20250905, rank]))
    vals = {k: v * (1 + 0.05 * rng.uniform(-1, 1, n_per_gpu)) for k, v in nominal.items()}
    return derive_batch(make, Fraction(1, 44100), vals, init_on_device=init_on_device)


def make_u(torch, dev, model, pots, amp, n, T, fs=FS):
    t = torch.arange(T, dtype=torch.float64, device=dev)
    sig = torch.sin(2 * np.pi * 1000.0 / fs * t)
    u = torch.empty((n, T, model.nu), dtype=torch.float64, device=dev)
    if np.isscalar(amp):
        u[:, :, 0] = amp * sig[None, :]
    else:
        u[:, :, 0] = torch.as_tensor(amp, dtype=torch.float64, device=dev)[:, None] * sig[None, :]
    if pots is not None:
        u[:, :, 1:] = torch.as_tensor(pots, dtype=torch.float64, device=dev)[:, None, :]
    return u


def pmc_record(workload, n, T, solver=None, kernel=None, kernel_ms=None):
    """The committed rocprofv3 PMC passes of THIS run's configuration (profiles/pmc_traffic.json, written by
    tools/profile_gpu.sh: separate --pmc runs of the same command), or None.  A record only counts if it was taken
    on the same workload, size, solver stack and kernel variant, and -- when this run's kernel time is given -- if
    the profiled launches lasted within 3 % of it: a stale number is worse than none."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            rec = json.load(fh)
    except OSError:
        return None
    for r in rec.get("runs", []):
        if r.get("workload") != workload or r.get("instances") != n or r.get("samples") != T:
            continue
        if solver is not None and r.get("solver") != solver:
            continue
        if kernel is not None and r.get("kernel") != kernel:
            continue
        prof = r.get("kernel_avg_ms_profiled")
        if kernel_ms is not None and (not prof or abs(prof - kernel_ms) > 0.03 * kernel_ms):
            continue
        return r
    return None


def kernel_name(runner, per_instance=False):
    """the dominant kernel as the rocprofv3 summaries name it (shape dimensions; condensed rows if any)"""
    nl, generic = runner.kernel_variant()
    if generic:
        return ("acme_coop_kernel" if runner.kernel_family() == "coop" else "acme_generic_kernel") + " (run-time dimensions %d,%d,%d,%d,%d,%d)" % runner.kernel_shape()
    shape = runner.kernel_shape()
    # (the two smallest shapes run one LANE per instance unless ACME_LANE_KERNEL=0 or the batch has private images:
    # csrc/acme_api.inc use_lane_kernel)
    lane = shape in ((2, 4, 1, 1, 1, 1), (2, 4, 2, 3, 1, 1)) and os.environ.get("ACME_LANE_KERNEL") != "0" and not per_instance
    name = ("acme_lane_kernel" if lane else "acme_run_kernel") + "<Shape<%d,%d,%d,%d,%d,%d" % shape
    return name + (", condensed rows %d>>" % nl if nl else ">>")


def pmc_traffic(r):
    """HBM bytes per launch from the PMC passes (record r of pmc_record, or None).  Reads = 2 x FETCH_SIZE: gfx950
    tallies this kernel's coalesced reads at half their size (calibrated on the known byte count of the u stream)."""
    if r is None:
        return None
    return 1024.0 * (2.0 * r["fetch_size_kb_per_launch"] + r["write_size_kb_per_launch"])


def pmc_valu_issue_frac(r, n_simd=1024, n_xcd=8):
    """Share of the chip's VALU issue slots the profiled launch used: SQ_INSTS_VALU wave-instructions
    against one 64-lane fp64 instruction per SIMD every 4 cycles (GRBM_GUI_ACTIVE sums the XCDs)."""
    if r is None or not r.get("sq_insts_valu_per_launch") or not r.get("grbm_gui_active_per_launch"):
        return None
    return r["sq_insts_valu_per_launch"] / (n_simd * r["grbm_gui_active_per_launch"] / n_xcd / 4.0)


def pmc_lds_bank_conflict_frac(r):
    """SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the profiled timed steps (north_star names the counter), or None."""
    if r is None or not r.get("sq_lds_idx_active_per_launch"):
        return None
    return r["sq_lds_bank_conflict_per_launch"] / r["sq_lds_idx_active_per_launch"]


def pmc_fp64_executed_flops(r):
    """fp64 operations the profiled launch EXECUTED: 64 lanes x (2 FMA + MUL + ADD + TRANS) wave-instructions
    (SQ_INSTS_VALU_*_F64).  Lanes masked off by EXEC are counted (the counters tally wave-instructions): an upper
    bound of the useful arithmetic, and the honest numerator for the FP64 pipe's utilisation."""
    if r is None or r.get("sq_insts_valu_fma_f64_per_launch") is None:
        return None
    return 64.0 * (2.0 * r["sq_insts_valu_fma_f64_per_launch"] + (r.get("sq_insts_valu_mul_f64_per_launch") or 0.0)
                   + (r.get("sq_insts_valu_add_f64_per_launch") or 0.0) + (r.get("sq_insts_valu_trans_f64_per_launch") or 0.0))


def pmc_exec_lane_frac(r):
    """mean share of a VALU instruction's 64 lanes that EXEC had switched on: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"""
    if r is None or not r.get("sq_thread_cycles_valu_per_launch") or not r.get("sq_active_inst_valu_per_launch"):
        return None
    return r["sq_thread_cycles_valu_per_launch"] / (64.0 * r["sq_active_inst_valu_per_launch"])


def lanes_with_rows_frac(model, runner):
    """Share of a 16-lane row's lanes that hold a row of the sub-problem in the 16-lane kernels (the others execute the
    same instructions on padding: structural, no counter sees it): (all rows / 16, rows in the Newton pass's elimination / 16
    -- the condensed shapes eliminate only the nonlinear rows there).  None for the lane-per-instance kernels."""
    if not model.subs or runner.kernel_family() == "generic":
        return None
    shape = runner.kernel_shape()
    nl, _ = runner.kernel_variant()
    nn = model.subs[0].nn
    if runner.kernel_family() == "coop":
        return (nn / (16.0 * -(-nn // 16)),) * 2
    if shape in ((2, 4, 1, 1, 1, 1), (2, 4, 2, 3, 1, 1)) and os.environ.get("ACME_LANE_KERNEL") != "0":
        return None
    return (nn / 16.0, (nn - nl) / 16.0)


def fp64_useful(executed_frac, r, rows):
    """fp64 utilisation with the idle lanes taken out: executed x EXEC-active share (counters) x share
    of the lanes that hold a row (structure).  An UPPER bound still: within the Newton pass of the condensed shapes only
    (nn - nl) / 16 of the lanes eliminate."""
    lf = pmc_exec_lane_frac(r)
    if executed_frac is None or lf is None or rows is None:
        return None
    return executed_frac * lf * rows[0]


def algorithmic_bytes(model, n, T):
    """8*(nu+ny) bytes per instance*sample + per-launch state/model traffic."""
    s = model.subs[0] if model.subs else None
    state = model.nx + (s.np + s.nn if s else 0)
    per_sample = 8 * (model.nu + model.ny)
    return n * T * per_sample + n * 2 * 8 * state


def algorithmic_flops(model, iters_per_sample):
    """SURVEY.md 8(d) flop model (sparse-aware), K = measured Newton evaluations/sample."""
    s = model.subs[0]
    nn, nq, np_, nx, nu, ny = s.nn, s.nq, s.np, model.nx, model.nu, model.ny
    nnz = sum({1: 2, 2: 6, 3: 6, 4: 3, 5: 2, 6: 4}[e["kind"]] for e in s.table)
    lu = 2 * nn ** 3 / 3
    fixed = 2 * np_ * (nx + nu) + 2 * nq * np_ + (np_ + 2 * nn * np_ + 2 * nn * nn + nn) + \
        2 * nnz * np_ + 2 * (ny + nx) * (nx + nu + nn)
    per_eval = 2 * nq * nn + 2 * nnz * nn + nn + lu + 2 * nn * nn + 2 * nn
    return fixed + per_eval * iters_per_sample


def _cpu_worker(args):
    """Times the oracle on `rows` (one fresh runner per stream): the first pass over the signal (cold
    solver state, empty solution cache) and, if `warm`, a second pass continuing it -- the regime the
    GPU's timed steps are in (bench.py warms the GPU runner up with the same signal first)."""
    fixture, rows, T, solver, warm, reflib = args
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
    from oracle import refpy
    from oracle.refpy import RefRunner
    if reflib:
        os.environ["ACME_REF_LIB"] = reflib
    else:
        os.environ.pop("ACME_REF_LIB", None)
    # the build that is about to be timed IS the one asked for (refpy caches per resolved path)
    want = os.path.realpath(reflib) if reflib else os.path.realpath(os.path.join(ROOT, "oracle", "libacme_ref.so"))
    assert refpy.lib().acme_path == want, (refpy.lib().acme_path, want)
    m = workload_model("", fixture, solver)
    cold = warm_t = 0.0
    iters = iters_warm = 0
    for u in rows:
        r = RefRunner(m)
        if solver == CachingHomotopySolver:
            r.set_cache_limit(16)      # the GPU's bounded store: same algorithm on both sides
        t0 = time.perf_counter()
        r.run(u)
        cold += time.perf_counter() - t0
        iters += r.report.iters_total
        if warm:
            t0 = time.perf_counter()
            r.run(u)
            warm_t += time.perf_counter() - t0
            iters_warm += r.report.iters_total
    return cold, iters, warm_t, iters_warm


def build_native_oracle():
    """-O3 -march=native build of oracle/acme_ref.c for THIS host (never shipped: not portable)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"libacme_ref_native_{os.getpid()}.so")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", f"NATIVE_OUT={out}"])
        return out
    except (OSError, subprocess.CalledProcessError):
        return None


def julia_probe(seconds=1.0, streams=4, timeout=900):
    """is there a `julia` on this box?  If so, time the REAL reference's run! (and push
    the doctest through julia/ACMEHip.jl) with julia/bench_reference.jl.  Returns None when Julia is absent
    (the case on every box seen so far), else a dict: the script's JSON line, or what went wrong."""
    import shutil
    import subprocess
    exe = shutil.which("julia")
    if exe is None:
        return None
    rec = {"julia": exe}
    try:
        out = subprocess.run([exe, "--startup-file=no", os.path.join(ROOT, "julia", "bench_reference.jl"), ROOT,
                              str(seconds), str(streams)], capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode == 0 and lines:
            rec.update(json.loads(lines[-1]))
        else:
            rec["error"] = (out.stderr or out.stdout)[-400:]
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        rec["error"] = repr(e)
    return rec


def host_cores():
    """CPUs this process may really use: min(affinity, cgroup quota) -- `nproc` on the GPU box
    reports the whole host (256) while the container is limited to a 16-CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(fixture, model, pots, amp, T_cpu, per_core=24, fs=FS, light=False):
    """Time the CPU oracle on a bounded, evenly spread sample of the same workload: one
    worker process per host core, `per_core` instance streams of T_cpu samples each (the
    reference's DiscreteModel is single-threaded and non-re-entrant, so independent per-core
    streams is how it would be scaled).  Throughput = units / slowest worker's compute time
    (process start-up and model loading are not charged to the CPU)."""
    import multiprocessing as mp
    from oracle import refpy
    refpy.lib()  # build once before forking
    cores = host_cores()
    n = len(pots) if pots is not None else (1 if np.isscalar(amp) else len(amp))   # scalar: identical streams
    pick = np.linspace(0, n - 1, cores * per_core).astype(int)
    sig = np.sin(2 * np.pi * 1000.0 / fs * np.arange(T_cpu))

    def jobs(warm, reflib, stride=1):
        out = []
        for w in range(cores):
            rows = []
            for i in pick[w::cores][::stride]:
                u = np.zeros((model.nu, T_cpu))
                u[0] = (amp if np.isscalar(amp) else amp[i]) * sig
                if pots is not None:
                    u[1:] = pots[i][:, None]
                rows.append(u)
            out.append((fixture, rows, T_cpu, model.solver, warm, reflib))
        return out
    with mp.get_context("fork").Pool(cores) as pool:
        # leg 1 (the headline CPU figure, as in round 1): -O2 build, cold streams; its second pass
        # over the same signal gives the warm-state figure
        res = pool.map(_cpu_worker, jobs(True, None), chunksize=1)
    # leg 2: the same source built -O3 -march=native on this host, on a third of the streams -- in FRESH
    # worker processes (and _cpu_worker asserts which build it has loaded).  (light: the other workloads' bounded
    # baselines -- config.other_workloads[*].cpu_baseline -- skip it: a few seconds each, not a minute)
    native = None if light else build_native_oracle()
    res_n = None
    if native:
        with mp.get_context("fork").Pool(cores) as pool:
            res_n = pool.map(_cpu_worker, jobs(True, native, stride=3), chunksize=1)
    if native:
        try:
            os.remove(native)
        except OSError:
            pass
    slowest = max(r[0] for r in res)
    units = len(pick) * T_cpu
    out = {
        "value": units / slowest, "unit": "circuit-instance*samples/sec", "cores": cores,
        "kind": "port",
        "sample": (f"{len(pick)} instances spread over the sweep" if n > 1 else
                   f"{len(pick)} streams of the nominal-component model (the Monte-Carlo instances differ in "
                   "component values, not in cost)") + f" x {T_cpu} samples, {per_core} oracle "
                  f"streams on each of {cores} cores, slowest worker {slowest:.1f} s "
                  "(C restatement oracle/acme_ref.c, gcc -O2, scalar; fresh solver state, as `value` of round 1)",
        "iters_per_sample": sum(r[1] for r in res) / units,
        # the same streams continued over a second pass of the signal: warm solver state and solution
        # caches, the regime of the GPU's timed steps
        "warm_value": units / max(r[2] for r in res),
        "warm_iters_per_sample": sum(r[3] for r in res) / units,
    }
    if res_n:
        units_n = sum(len(j[1]) for j in jobs(False, None, stride=3)) * T_cpu
        out["native_value"] = units_n / max(r[0] for r in res_n)
        out["native_warm_value"] = units_n / max(r[2] for r in res_n)
        out["native_note"] = "gcc -O3 -march=native build of the same C source on this host, a third of the streams"
    return out


def other_workload_leg(workload, local_rank, dev, steps=2, warmup=3, with_cpu=True, n=None, T=None):
    """One short steady-state measurement of another BASELINE configuration (config.other_workloads; never `value`):
    the same procedure as the headline's timed steps -- fresh batch, `warmup` launches continuing into `steps` timed
    ones -- at the configuration's own size and solver stack."""
    import torch
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel, HomotopySolver
    from acme_jl_amd.runner import ModelRunner
    n = n or {"diodeclipper_sweep": 4096, "birdie_grid": 2048}.get(workload, 8192)
    fs = 176400 if workload == "birdie_grid" else FS
    T = T or (fs // 10 if workload.startswith("clipper_chain") else fs)
    solver = HomotopySolver if workload == "birdie_grid" else CachingHomotopySolver
    fixture, pots, amp = grid_inputs(workload, 0, 1, n, T)
    model = workload_model(workload, fixture, solver, fs)
    if workload == "superover_montecarlo":
        batch = montecarlo_models(0, n, init_on_device={"device": local_rank})
        batch.solver = solver
        model = batch.model(0)
        runner = ModelRunner(model, n, device=local_rank, models=batch)
    else:
        runner = ModelRunner(model, n, device=local_rank)
    u = make_u(torch, dev, model, pots, amp, n, T, fs)
    y = torch.empty((n, T, model.ny), dtype=torch.float64, device=dev)
    # (a leg that follows a CPU-baseline leg finds the GPU at idle clocks, and three short warm-up steps do not always bring
    # them back: half a second of unrelated work first -- the warm-up steps themselves stay what the profiles' are)
    spin = torch.ones((2048, 2048), dtype=torch.float64, device=dev)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.5:
        spin = (spin @ spin) * 0.0 + 1.0
        torch.cuda.synchronize()
    del spin
    for _ in range(warmup):
        runner.run_torch(u, y)
    torch.cuda.synchronize()
    runner.reset_report()
    runner.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.run_torch(u, y)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms_total, launches = runner.kernel_time()
    ra = runner.report_arrays()
    kname, kms = kernel_name(runner, workload == "superover_montecarlo"), ms_total / max(launches, 1)
    prec = pmc_record(workload, n, T, model.solver, kname, kms)       # (this configuration's own PMC pass, or nothing)
    fx = pmc_fp64_executed_flops(prec)
    abytes = algorithmic_bytes(model, n, T)
    roof = {"achieved": abytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": abytes, "traffic": pmc_traffic(prec), "valu_issue_frac": pmc_valu_issue_frac(prec),
            "lds_bank_conflict_frac": pmc_lds_bank_conflict_frac(prec),
            "fp64_executed_tflops": (fx / (prec["kernel_avg_ms_profiled"] * 1e-3) / 1e12) if fx else None,
            "exec_active_lane_frac": pmc_exec_lane_frac(prec), "lanes_with_rows_frac": lanes_with_rows_frac(model, runner),
            "fp64_useful_frac": fp64_useful((fx / (prec["kernel_avg_ms_profiled"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if fx else None, prec,
                                            lanes_with_rows_frac(model, runner)),
            "wave_wait_frac": (prec["sq_wait_any_per_launch"] / prec["sq_wave_cycles_per_launch"])
            if prec and prec.get("sq_wait_any_per_launch") is not None and prec.get("sq_wave_cycles_per_launch") else None,
            "profiled_kernel_ms": (prec or {}).get("kernel_avg_ms_profiled")}
    cpu = None
    if with_cpu:
        # the CPU oracle on a bounded sample of the SAME synthetic inputs (a tenth of a second of signal, four
        # streams per core spread over the sweep: seconds, so that the default run still ends within minutes)
        try:
            cpu = cpu_baseline(fixture, model, pots, amp, fs // 10, per_core=4, fs=fs, light=True)
        except Exception as e:
            cpu = {"error": repr(e)}
    return {"workload": workload, "roofline": roof, "config": {"diodeclipper_sweep": 2, "superover_montecarlo": 4, "birdie_grid": 5}.get(workload),
            "cpu_baseline": cpu,
            "waves_per_simd": waves_per_simd(runner, n),
            "instances": n, "samples_per_step": T, "fs": fs, "solver": model.solver, "steps": steps, "warmup": warmup,
            "value": n * T * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "kernel_ms": kms,
            "kernel": kname,
            "newton_iters_per_sample": float(ra["iters_total"].sum()) / (n * T * steps), "n_warn": float(ra["n_warn"].sum()),
            "y_abs_sum": float(torch.nan_to_num(y).abs().sum())}


def mid_size_leg(local_rank, dev, sizes=(24, 32, 34, 48, 64), n=8192, steps=2, warmup=2):
    """config.mid_size (never `value`): the mid-size kernel (csrc/acme_coop.h) over the range the reference's LU is written for
    (src/solvers.jl:53-54, "sizes up to about 60 x 60") -- the clipper chain of clipper_chain_20 with 12 ... 32 stages, ONE
    undecomposed sub-problem of 24 ... 64 unknowns, 8 192 instances, a twentieth of a second per step: 17 ... 32 unknowns
    run with the Jacobian's rows in registers, 33 ... 64 on one matrix per instance in LDS with one instance per wave."""
    import torch
    from fractions import Fraction
    from acme_jl_amd import examples
    from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
    from acme_jl_amd.runner import ModelRunner
    T = FS // 20
    amp = 10.0 ** (-2 + 2.7 * np.arange(n) / max(n - 1, 1))
    out = []
    for nn in sizes:
        try:
            model = DiscreteModel(examples.clipper_chain(nn // 2), Fraction(1, FS), CachingHomotopySolver, decompose_nonlinearity=False)
            runner = ModelRunner(model, n, device=local_rank)
            u = make_u(torch, dev, model, None, amp, n, T)
            y = torch.empty((n, T, model.ny), dtype=torch.float64, device=dev)
            for _ in range(warmup):
                runner.run_torch(u, y)
            torch.cuda.synchronize()
            runner.reset_report()
            runner.kernel_time(reset=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                runner.run_torch(u, y)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            ms_total, launches = runner.kernel_time()
            ra = runner.report_arrays()
            out.append({"unknowns": nn, "instances": n, "samples_per_step": T, "steps": steps, "warmup": warmup, "kernel_family": runner.kernel_family(),
                        "value": n * T * steps / elapsed, "kernel_ms": ms_total / max(launches, 1),
                        "newton_iters_per_sample": float(ra["iters_total"].sum()) / (n * T * steps), "n_warn": float(ra["n_warn"].sum()),
                        "y_abs_sum": float(torch.nan_to_num(y).abs().sum())})
            del runner, u, y
        except Exception as e:      # (the headline line must come out whatever happens here)
            out.append({"unknowns": nn, "error": repr(e)})
    return out


def waves_per_simd(runner, n, n_simd=1024):
    """resident wavefronts per SIMD this batch puts on the chip (256 CUs x 4 SIMDs): 16 lanes per instance in the tuned and
    the mid-size kernels (4 instances per wave), one lane per instance in the generic kernel; the lane-per-instance kernel
    of the two smallest shapes spreads its instances thin (csrc/acme_lane_kernel.h: 4 ... 64 per wave, the launcher's
    choice) and is reported as a range"""
    fam = runner.kernel_family()
    if fam == "generic":
        return -(-n // 64) / n_simd
    shape = runner.kernel_shape()
    if fam == "tuned" and shape in ((2, 4, 1, 1, 1, 1), (2, 4, 2, 3, 1, 1)) and os.environ.get("ACME_LANE_KERNEL") != "0":
        return [-(-n // 64) / n_simd, -(-n // 4) / n_simd]
    return -(-n // 4) / n_simd


def literal_grid_leg(local_rank, dev, T=160):
    """SURVEY 8(d)'s LITERAL config-3 grid -- drive in linspace(0, 1, 32) exactly (test/runtests.jl:778) -- which `value`
    does not run: its last column (256 of 8 192 instances) sits on the singular drive = 1.0 corner of the variable-pot
    model, where the reference's solver stack fails after ~900 Newton iterations on practically every sample and warns
    (src/ACME.jl:688-690) -- a launch lasts as long as its slowest wave.  With acme_batch_set_isolation the library runs
    the instances it has seen misbehave in a launch of their own: reported are the rate at which the 7 936 healthy
    instances complete on the caller's stream, the whole grid's rate, and the singular column's warnings.  A few hundred
    samples only: a singular cell costs ~50 ms of GPU time per sample."""
    import torch
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    N = 8192
    model = workload_model("superover_grid", "superover_var", CachingHomotopySolver)
    idx = np.arange(N)
    pots = np.stack([(idx // 256) / 31.0, ((idx // 16) % 16) / 15.0, (idx % 16) / 15.0], axis=1)
    healthy = pots[:, 0] < 1.0
    u = make_u(torch, dev, model, pots, 1.0, N, T)
    r = ModelRunner(model, N, device=local_rank)
    r.set_isolation(50.0)
    out = {"workload": "superover_grid_literal", "instances": N, "samples_per_step": T, "solver": model.solver,
           "singular_instances": int((~healthy).sum())}
    for step in ("classifying", "isolated"):       # the first run measures who is slow, the second runs them apart
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = r.run_torch(u)
        torch.cuda.current_stream().synchronize()
        t_stream = time.perf_counter() - t0
        r.wait(check=False)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        out[step + "_ms_callers_stream"] = 1e3 * t_stream
        out[step + "_ms_everything"] = 1e3 * t_all
    ra = {k: np.asarray(v) for k, v in r.report_arrays().items()}
    hm = healthy
    out.update(value_healthy_instances=int(healthy.sum()) * T / t_stream, value_whole_grid=N * T / t_all,
               n_warn_singular_column=float(ra["n_warn"][~hm].sum()), n_warn_healthy=float(ra["n_warn"][hm].sum()),
               samples_of_the_singular_column=int((~healthy).sum()) * T * 2,
               newton_iters_per_sample_singular=float(ra["iters_total"][~hm].sum()) / (int((~healthy).sum()) * T * 2),
               newton_iters_per_sample_healthy=float(ra["iters_total"][hm].sum()) / (int(healthy.sum()) * T * 2),
               y_finite=bool(torch.isfinite(y).all()),
               note="drive = linspace(0, 1, 32): value_healthy_instances = the 7 936 instances with drive < 1 x samples / time until "
                    "the caller's stream is done (second run: acme_batch_set_isolation(50) has moved the singular column to a "
                    "launch of its own); the reference warns on (practically) every sample of the singular column and carries on "
                    "(src/ACME.jl:688-690), as this library does: n_warn_singular_column of samples_of_the_singular_column")
    return out


def saturation_curve(local_rank, dev, factors=(1, 2, 4, 8)):
    """--saturation: every workload at 1 x, 2 x, 4 x, 8 x its BASELINE per-GPU instance count (config 5 at 8 x is ALL of its
    16 384 instances on one GPU), a quarter of a second of signal per step, steady state: where one chip saturates, and
    hence when sharding a sweep over more GPUs stops paying.  config.saturation; never `value`."""
    import torch
    rows = []
    for wl, base in (("superover_grid", 8192), ("diodeclipper_sweep", 4096), ("superover_montecarlo", 8192), ("birdie_grid", 2048),
                     ("clipper_chain_20", 8192)):
        fs = 176400 if wl == "birdie_grid" else FS
        for f in factors:
            try:
                r = other_workload_leg(wl, local_rank, dev, steps=2, warmup=2, with_cpu=False, n=base * f,
                                       T=fs // 10 if wl == "clipper_chain_20" else fs // 4)
                rows.append({k: r[k] for k in ("workload", "config", "instances", "samples_per_step", "value", "ms_per_step", "kernel_ms",
                                               "newton_iters_per_sample", "waves_per_simd")} | {"factor": f})
            except Exception as e:
                rows.append({"workload": wl, "factor": f, "instances": base * f, "error": repr(e)[:200]})
            torch.cuda.empty_cache()
    return rows


def host_buffer_leg(runner, u, N, T, model):
    """run! through host buffers (see main): instance*samples/s of the first call (page-locks u and y), of the
    following calls on the same arrays (steady state) and, on a second pair of arrays, from pageable memory."""
    import ctypes as C
    from acme_jl_amd.runner import ACME_MEM_HOST
    dp = C.POINTER(C.c_double)
    uh = u.cpu().numpy()                       # [N][T][nu]: the ABI's (and Julia's nu x T x N) layout
    yh = np.zeros((N, T, model.ny))
    yh.fill(0.0)                               # (pages touched: the first call below is the library's cost, not the page faults
                                               #  of a fresh 2.9 GB allocation, which any caller's first write pays)

    def call(ub, yb):
        t0 = time.perf_counter()
        runner.lib.check(runner.lib.L.acme_batch_run(runner.h, ub.ctypes.data_as(dp), yb.ctypes.data_as(dp), T, ACME_MEM_HOST, None))
        return time.perf_counter() - t0
    out = {"bytes_in": int(uh.nbytes), "bytes_out": int(yh.nbytes)}
    os.environ.pop("ACME_HOST_REGISTER", None)
    # what a caller gets who promises nothing (the default; what ModelRunner.run and Julia's run! on fresh arrays do):
    # ONE call, nothing page-locked, nothing of the arrays kept
    t_oneshot = call(uh, yh)
    out.update(one_shot_ms=1e3 * t_oneshot, one_shot_value=N * T / t_oneshot)
    # ... and a caller who reuses its arrays and says so (acme_batch_set_host_retention)
    runner.set_host_retention(True)
    t_first = call(uh, yh)
    ts = [call(uh, yh) for _ in range(2)]
    out.update(first_call_ms=1e3 * t_first, steady_ms=1e3 * min(ts), steady_value=N * T / min(ts),
               first_call_value=N * T / t_first, y_abs_sum=float(np.abs(np.nan_to_num(yh)).sum()))
    extra = {}
    for key, env in (("inplace", {"ACME_HOST_SLICES": "1"}), ("staged", {"ACME_HOST_ZEROCOPY": "0"})):
        os.environ.update(env)
        try:
            extra[key] = min(call(uh, yh) for _ in range(2))
        finally:
            for k in env:
                os.environ.pop(k, None)
    if os.environ.get("ACME_BENCH_HOST_SLICES"):        # (developer: sweep of the default pipeline's slice count)
        for ns in os.environ["ACME_BENCH_HOST_SLICES"].split(","):
            os.environ["ACME_HOST_SLICES"] = ns
            try:
                out["slices_%s_ms" % ns] = 1e3 * min(call(uh, yh) for _ in range(2))
            finally:
                os.environ.pop("ACME_HOST_SLICES", None)
    runner.release_host_buffers()
    runner.set_host_retention(False)
    os.environ["ACME_HOST_REGISTER"] = "0"
    try:
        t_page = min(call(uh, yh) for _ in range(2))
    finally:
        os.environ.pop("ACME_HOST_REGISTER", None)
    # constant input rows (acme_batch_run_const): rows that never change during the call -- the three pot positions of the
    # headline grid -- handed over ONCE per instance; a one-shot call from pageable memory then moves a quarter of the bytes
    const_rows = [k for k in range(model.nu) if np.all(uh[:, :, k] == uh[:, :1, k])]
    if const_rows and len(const_rows) < model.nu:
        var_rows = [k for k in range(model.nu) if k not in const_rows]
        uv, uc = np.ascontiguousarray(uh[:, :, var_rows]), np.ascontiguousarray(uh[:, 0, :])
        mask = sum(1 << k for k in const_rows)
        yc = np.zeros_like(yh)

        def call_const():
            t0 = time.perf_counter()
            runner.lib.check(runner.lib.L.acme_batch_run_const(runner.h, uv.ctypes.data_as(dp), uc.ctypes.data_as(dp), mask,
                                                               yc.ctypes.data_as(dp), T, ACME_MEM_HOST, None))
            return time.perf_counter() - t0
        t_const = min(call_const() for _ in range(2))
        out.update(const_rows=const_rows, const_rows_ms=1e3 * t_const, const_rows_value=N * T / t_const, const_rows_bytes_in=int(uv.nbytes + uc.nbytes),
                   const_rows_y_abs_sum=float(np.abs(np.nan_to_num(yc)).sum()))
        # ... and for a caller who keeps and reuses its arrays (acme_batch_set_host_retention: u_var and y page-locked once)
        runner.set_host_retention(True)
        try:
            t_lock = call_const()
            t_kept = min(call_const() for _ in range(2))
            out.update(const_rows_retained_first_ms=1e3 * t_lock, const_rows_retained_ms=1e3 * t_kept, const_rows_retained_value=N * T / t_kept)
        finally:
            runner.release_host_buffers()
            runner.set_host_retention(False)
    out.update(inplace_ms=1e3 * extra["inplace"], inplace_value=N * T / extra["inplace"],
               staged_ms=1e3 * extra["staged"], staged_value=N * T / extra["staged"],
               pageable_ms=1e3 * t_page, pageable_value=N * T / t_page,
               note="acme_batch_run(ACME_MEM_HOST) on the timed batch, continuing the signal.  const_rows_*: acme_batch_run_const from pageable "
                    "memory -- the input rows that are constant over the call (const_rows) handed over once per instance, the varying "
                    "rows as [N][T][nu_var], the full rows put together on the device; compare with one_shot / pageable.  "
                    "one_shot: the default -- a single "
                    "call on arrays the library may not keep anything of (pageable memory, 24 time slices staged through HBM "
                    "with the copies overlapping the kernel).  The rest with acme_batch_set_host_retention (the caller keeps "
                    "its arrays alive and reuses them).  first call: page-locks the "
                    "caller's arrays; steady: the same arrays again -- the STREAMED pipeline: one launch over the whole "
                    "run, y written to the locked host array by the kernel itself, u copied into an HBM staging buffer by "
                    "the copy engine 128 samples of every row at a time while the kernel runs (a wave that gets ahead of "
                    "the copy waits: KArgs::u_ready); inplace (ACME_HOST_SLICES=1): one launch, u and y both in place over "
                    "the bus; staged (ACME_HOST_ZEROCOPY=0): "
                    "24 time slices, u and y both through HBM, copies overlapped with the kernel on two streams; pageable "
                    "(ACME_HOST_REGISTER=0): the staged pipeline from unlocked memory")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults = the driver's command: the first steps of a fresh batch learn their solution caches and have no measured
    # wave costs to place by -- a step is 0.27 s, the whole default run about a minute and a half)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="superover_grid",
                    choices=["superover_grid", "diodeclipper_sweep", "superover_montecarlo", "birdie_grid", "clipper_chain_20", "clipper_chain_34"],
                    help="superover_grid = BASELINE config 3 (the headline, default); diodeclipper_sweep = "
                         "config 2; superover_montecarlo = config 4 (per-instance model blocks); birdie_grid = "
                         "config 5 (176.4 kHz, 2048 instances per GPU, HomotopySolver unless --solver is given); "
                         "clipper_chain_20 / _34 = beyond BASELINE, one 20- / 34-unknown sub-problem on the mid-size kernel (rows in registers / "
                         "matrix in LDS, one instance per wave), 0.1 s per step")
    ap.add_argument("--instances", type=int, default=None, help="instances per GPU")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short steady-state legs of BASELINE configs 2, 4, 5 (config.other_workloads) that follow the headline")
    ap.add_argument("--samples", type=int, default=None, help="samples per step (default: 1 s of signal)")
    ap.add_argument("--solver", default=None, choices=["caching", "homotopy", "simple"],
                    help="caching = HomotopySolver{CachingSolver{SimpleSolver}}, the reference's default stack "
                         "(GPU: bounded 16-entry store per instance); homotopy = HomotopySolver{SimpleSolver}")
    ap.add_argument("--gather", default=None, choices=["none", "rank0", "allgather"],
                    help="collection of the sharded outputs after the timed steps, timed separately (gather_ms): "
                         "rank0 = every rank sends its y to rank 0 (grouped send/recv), allgather = RCCL all-gather; "
                         "default: rank0 when N > 1, none at N = 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the host-buffer leg (config.value_host_buffers: run! through acme_batch_run(ACME_MEM_HOST), "
                         "what the Julia binding calls)")
    ap.add_argument("--cpu-samples", type=int, default=None)
    ap.add_argument("--saturation", action="store_true",
                    help="also measure every workload at 1 / 2 / 4 / 8 x its BASELINE per-GPU instance count (config.saturation; "
                         "several minutes; all 16 384 instances of config 5 on one GPU among them)")
    ap.add_argument("--no-literal-grid", action="store_true",
                    help="skip the leg on the LITERAL config-3 grid, drive = linspace(0, 1, 32) with its singular column (config.literal_grid)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # ACME_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank code path on a single-GPU box (every rank
    # on cuda:0, gloo instead of RCCL); never set by the driver
    rehearsal = os.environ.get("ACME_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ACME_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, broadcast, all-reduce,
    # barriers) even at world size 1 -- what the single-GPU box can check of it (tests)
    use_dist = world > 1 or os.environ.get("ACME_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from acme_jl_amd.dist import broadcast_model
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.runner import ModelRunner

    n_per_gpu = args.instances or {"diodeclipper_sweep": 4096, "birdie_grid": 2048}.get(args.workload, 8192)
    fs = 176400 if args.workload == "birdie_grid" else FS
    T = args.samples or (fs // 10 if args.workload.startswith("clipper_chain") else fs)
    args.solver = args.solver or ("homotopy" if args.workload == "birdie_grid" else "caching")
    fixture, pots, amp = grid_inputs(args.workload, rank, world, n_per_gpu, T)
    # rank 0 owns the model block; everyone else receives it over RCCL (xGMI)
    from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver, SimpleSolver
    solver = {"caching": CachingHomotopySolver, "homotopy": HomotopySolver, "simple": SimpleSolver}[args.solver]
    model = workload_model(args.workload, fixture, solver, fs) if rank == 0 else None
    model = broadcast_model(model, src=0, device=dev) if use_dist else model

    setup = {}
    if args.workload == "superover_montecarlo":
        t0 = time.perf_counter()
        init_info = {"device": local_rank}       # initial_solution of every instance: one batched GPU solve
        batch = montecarlo_models(rank, n_per_gpu, init_on_device=init_info)
        setup["derive_s"] = time.perf_counter() - t0
        setup["initial_solutions_on_gpu"] = init_info.get("solved", 0)
        t0 = time.perf_counter()
        batch.solver = solver
        model = batch.model(0)
        runner = ModelRunner(model, n_per_gpu, device=local_rank, models=batch)
        setup["upload_s"] = time.perf_counter() - t0
    else:
        runner = ModelRunner(model, n_per_gpu, device=local_rank)
    u = make_u(torch, dev, model, pots, amp, n_per_gpu, T, fs)
    y = torch.empty((n_per_gpu, T, model.ny), dtype=torch.float64, device=dev)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    cold_ms = None
    for w in range(args.warmup):
        if w == 0:      # the first step of a fresh batch: cold solver state, empty solution caches
            sync()
            tc = time.perf_counter()
        runner.run_torch(u, y)
        if w == 0:
            sync()
            cold_ms = 1e3 * (time.perf_counter() - tc)
    sync()
    runner.reset_report()
    runner.kernel_time(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.run_torch(u, y)
    sync()
    elapsed = time.perf_counter() - t0
    # average launch duration over the timed region: HIP events recorded by the library on
    # the launch stream around every kernel
    ms_total, launches = runner.kernel_time()
    last_ms = ms_total / max(launches, 1)
    ra = runner.report_arrays()
    if os.environ.get("ACME_BENCH_DUMP_ITERS"):      # per-instance iteration totals of the timed steps (tools/blockload.py)
        np.save(os.environ["ACME_BENCH_DUMP_ITERS"], np.asarray(ra["iters_total"].cpu() if hasattr(ra["iters_total"], "cpu") else ra["iters_total"]))
    stats = torch.tensor([elapsed, float(ra["iters_total"].sum()), float(ra["n_warn"].sum()),
                          float((ra["first_nonfinite"] >= 0).sum()), float(ra["iters_max"].max()),
                          last_ms], dtype=torch.float64, device=dev)
    if use_dist:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        iters_total, n_warn, n_dead = float(sm[1]), float(sm[2]), float(sm[3])
        iters_max, last_ms = float(mx[4]), float(mx[5])
    else:
        iters_total, n_warn, n_dead, iters_max = (float(stats[1]), float(stats[2]), float(stats[3]),
                                                  float(stats[4]))
    checksum = float(torch.nan_to_num(y).abs().sum())
    checksums = [checksum]
    if use_dist:       # every rank's output checksum (shards must differ: other grid cells / other seeds)
        cs = [None] * world
        dist.all_gather_object(cs, checksum)
        checksums = [float(c) for c in cs]
    # Collection of the sharded outputs (SURVEY 8e / north star: "gather of outputs over xGMI"): the
    # last step's y of every rank, timed on its own between barriers.  It is not part of `value`
    # (instances never interact: no collective sits on the data path); `value_incl_gather` charges
    # one such collection to every step.
    gather_mode = args.gather or ("rank0" if use_dist and world > 1 else "none")
    if os.environ.get("ACME_BENCH_FORCE_DIST") == "1" and args.gather is None:
        gather_mode = "rank0"          # single-rank rehearsal of the collective path (tests)
    gather_ms = gathered_shape = None
    if gather_mode != "none" and use_dist:
        from acme_jl_amd.dist import collect_outputs
        sync()
        tg = time.perf_counter()
        full = collect_outputs(y, mode=gather_mode, dst=0)
        sync()
        gather_ms = 1e3 * (time.perf_counter() - tg)
        gm = torch.tensor([gather_ms], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        gather_ms = float(gm[0])
        if full is not None:
            gathered_shape = list(full.shape)
            assert gathered_shape == [world * n_per_gpu, T, model.ny]
            assert torch.equal(full[rank * n_per_gpu:(rank + 1) * n_per_gpu].to(y.device), y)
        del full

    # The host-buffer path -- acme_batch_run(ACME_MEM_HOST), what julia/ACMEHip.jl's run! calls with the caller's
    # arrays (src/ACME.jl:650-664) -- on the same batch, continuing the signal (steady state): u and y in host memory,
    # in time slices whose copies overlap the kernel.  First call: page-locks the arrays; steady: the same arrays again
    # (a caller reusing its arrays); pageable: ACME_HOST_REGISTER=0.  Never `value`.
    host = None
    if world == 1 and not args.no_host_path:
        host = host_buffer_leg(runner, u, n_per_gpu, T, model)
    # BASELINE configs 2, 4 and 5 next to the headline (config 3): one short steady-state measurement each, reported
    # under config.other_workloads -- every BASELINE number in the driver's own record -- and, beyond BASELINE, one
    # 20-unknown sub-problem on the mid-size kernel (config: null).  Never `value`.
    others = None
    if world == 1 and args.workload == "superover_grid" and not args.no_other_workloads:
        others = []
        for wl in ("diodeclipper_sweep", "superover_montecarlo", "birdie_grid", "clipper_chain_20", "clipper_chain_34"):
            try:
                others.append(other_workload_leg(wl, local_rank, dev))
            except Exception as e:      # (the headline line must come out whatever happens here)
                others.append({"workload": wl, "error": repr(e)})
    mid = None
    if world == 1 and args.workload == "superover_grid" and not args.no_other_workloads:
        mid = mid_size_leg(local_rank, dev)
    literal = saturation = None
    if world == 1 and args.workload == "superover_grid" and not args.no_other_workloads and not args.no_literal_grid:
        try:
            literal = literal_grid_leg(local_rank, dev)
        except Exception as e:
            literal = {"workload": "superover_grid_literal", "error": repr(e)}
    if world == 1 and args.saturation:
        saturation = saturation_curve(local_rank, dev)
    if rank == 0:
        units = world * n_per_gpu * T * args.steps
        value = units / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        iters_per_sample = iters_total / units
        abytes = algorithmic_bytes(model, n_per_gpu, T)
        achieved = abytes / (last_ms * 1e-3) / 1e9
        kname = kernel_name(runner, args.workload == "superover_montecarlo")
        prec = pmc_record(args.workload, n_per_gpu, T, model.solver, kname, last_ms)
        fx = pmc_fp64_executed_flops(prec)
        out = {
            "metric": {"diodeclipper_sweep": "circuit-instance*samples/sec (diodeclipper, 44.1 kHz)",
                       "birdie_grid": "circuit-instance*samples/sec (birdie, 176.4 kHz)"}.get(
                           args.workload, "circuit-instance*samples/sec (superover, 44.1 kHz)"),
            "value": value, "unit": "circuit-instance*samples/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("examples/superover.jl (pots as inputs: nn=13,nq=29,np=11,nx=11,nu=4), "
                             f"{n_per_gpu}-instance drive x tone x level grid per GPU "
                             f"(drive=i/{n_per_gpu * world // 256}, tone,level=linspace(0,1,16)), "
                             "1 kHz unit sine")
                if args.workload == "superover_grid" else
                (f"examples/superover.jl fixed pots (nn=7,nq=14,np=5,nx=11,nu=1), {n_per_gpu} Monte-Carlo "
                 "instances per GPU: R, C, pot tracks * (1 + 0.05 U(-1,1)), PCG64 seed 20250905, one model "
                 "block per instance; 1 kHz unit sine")
                if args.workload == "superover_montecarlo" else
                (f"examples/birdie.jl at 176.4 kHz, vol as input (nn=4,nq=9,np=3,nx=3,nu=2), {n_per_gpu} instances "
                 f"per GPU: {n_per_gpu * world // 128} amplitudes 10^(-2..0.5) x 128 vol in linspace(0.01,1), "
                 "1 kHz sine")
                if args.workload == "birdie_grid" else
                (f"beyond BASELINE: a chain of {model.subs[0].nn // 2} diode-clipper stages, undecomposed = ONE nonlinear sub-problem "
                 f"(nn={model.subs[0].nn},nq={model.subs[0].nq},np={model.subs[0].np},nx={model.nx},nu=1; acme_jl_amd.examples.clipper_chain), "
                 f"{n_per_gpu}-instance amplitude sweep 10mV..5V per GPU, 1 kHz sine: the cooperative mid-size kernel")
                if args.workload.startswith("clipper_chain") else
                f"examples/diodeclipper.jl, {n_per_gpu}-instance amplitude sweep 10mV..10V per GPU",
                "instances_per_gpu": n_per_gpu, "samples_per_step": T, "fs": fs,
                "solver": model.solver,
                "solver_note": "caching = the reference's default stack; GPU and CPU oracle keep the last 16 stored "
                               "solutions per instance (reference: unbounded k-d tree), same lookup/store rules; "
                               "--solver homotopy runs HomotopySolver{SimpleSolver}",
                "parallelism": f"instance-sharded x{world}",
                "rccl_ranks": world if (use_dist and not rehearsal) else 0,
                "collective_backend": ("gloo (one-device rehearsal)" if rehearsal else "nccl (RCCL)") if use_dist else None,
                "gather": gather_mode, "gather_ms": gather_ms, "gathered_shape": gathered_shape,
                "value_incl_gather": (units / (elapsed + args.steps * gather_ms * 1e-3)) if gather_ms is not None else None,
                "cold_first_step_ms": cold_ms,
                "value_host_buffers": host["steady_value"] if host else None,
                "host_buffers": host,
                "other_workloads": others,
                "mid_size": mid,
                "literal_grid": literal,
                "saturation": saturation,
                "grid_note": "`value` runs drive = i/32 (i = 0 ... 31): SURVEY 8(d) quotes linspace(0, 1, 32), whose last column "
                             "(drive = 1.0) is a singular corner of the variable-pot model where the reference warns on every sample; "
                             "config.literal_grid has that grid, with acme_batch_set_isolation",
                "timed_steps_note": "the timed steps continue the signal of the warm-up steps: warm solver state and "
                                    "solution caches (steady state); cold_first_step_ms is the first step of the fresh batch",
                "newton_iters_per_sample": iters_per_sample, "iters_max": iters_max,
                "n_warn": n_warn, "n_nonfinite_instances": n_dead, "y_abs_sum_rank0": checksum,
                "y_abs_sum_per_rank": checksums,
                **({"host_setup": setup} if setup else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(prec),
                "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
                "kernel": kname,
                "kernel_ms": last_ms, "algorithmic_bytes_per_launch": abytes,
                "note": "path is bound by per-wave instruction issue/fetch, not by HBM (DESIGN.md 2).  Two fp64 figures: "
                        "fp64_executed_* = the fp64 instructions the kernel executed (PMC SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64 "
                        "lanes, FMA = 2 flop; masked lanes included) -- the pipe's utilisation; fp64_reference_equivalent_* = the "
                        "flop model of SURVEY 8(d) at the measured Newton iterations per sample -- the work the REFERENCE's "
                        "algorithm does for the same samples: it counts a dense 13 x 13 LU and 29-row q products per iteration, "
                        "which the condensed kernel does not perform (7-step elimination on the reduced system), so it is a "
                        "work-equivalent rate, not utilisation",
                "fp64_executed_tflops": (fx / (prec["kernel_avg_ms_profiled"] * 1e-3) / 1e12) if fx else None,
                "fp64_executed_frac": (fx / (prec["kernel_avg_ms_profiled"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if fx else None,
                "exec_active_lane_frac": pmc_exec_lane_frac(prec),
                "lanes_with_rows_frac": lanes_with_rows_frac(model, runner),
                "fp64_useful_frac": fp64_useful((fx / (prec["kernel_avg_ms_profiled"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if fx else None, prec,
                                                lanes_with_rows_frac(model, runner)),
                "fp64_useful_note": "fp64_executed_frac x exec_active_lane_frac (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU: lanes EXEC "
                                    "had switched on) x lanes_with_rows_frac[0] (lanes of a 16-lane row that hold a row of the sub-problem; "
                                    "[1]: the share that eliminates in the Newton pass of a condensed shape): the kernels predicate with "
                                    "selects, not with EXEC, so the counters see nearly all lanes active and the structural factor is what "
                                    "takes the padding lanes out -- an upper bound of useful fp64",
                "fp64_reference_equivalent_tflops": algorithmic_flops(model, iters_per_sample) * n_per_gpu * T
                / (last_ms * 1e-3) / 1e12 if model.subs else None,
                "fp64_peak_tflops": FP64_PEAK_TFLOPS,
                "lds_bank_conflict_frac": pmc_lds_bank_conflict_frac(prec),
                "valu_issue_frac": pmc_valu_issue_frac(prec),
                "valu_issue_profiled_kernel_ms": (prec or {}).get("kernel_avg_ms_profiled"),
                "valu_issue_note": "VALU wave-instructions issued / (1024 SIMDs x cycles / 4) and LDS bank-conflict cycles / "
                                   "LDS-active cycles, from the committed rocprofv3 PMC passes of this workload, solver stack and "
                                   "kernel variant (profiles/pmc_traffic.json, tools/profile_gpu.sh: means over the dispatches of the "
                                   "TIMED steps only, whose mean duration is valu_issue_profiled_kernel_ms and must lie within 3 % of "
                                   "this run's kernel_ms); null if no such record is on file",
            },
        }
        if use_dist and not rehearsal:
            assert out["config"]["rccl_ranks"] == args.gpus == dist.get_world_size(), (world, args.gpus)
            assert dist.get_backend() == "nccl"
            assert len(set(checksums)) == world, f"ranks computed identical shards: {checksums}"
        if out["roofline"]["fp64_reference_equivalent_tflops"] is not None:
            out["roofline"]["fp64_reference_equivalent_frac"] = out["roofline"]["fp64_reference_equivalent_tflops"] / FP64_PEAK_TFLOPS
        if world == 1 and not args.no_cpu_baseline:
            T_cpu = args.cpu_samples or min(T, FS)
            out["cpu_baseline"] = cpu_baseline(fixture, model, pots, amp, T_cpu, fs=fs)
            # the real reference, if this box has a Julia (null: `julia` not on PATH)
            out["cpu_baseline"]["julia"] = julia_probe()
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
