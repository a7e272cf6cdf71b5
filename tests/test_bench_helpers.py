"""CPU-side checks of bench.py's workload definitions (the timed part needs a GPU)."""
import numpy as np

import bench


def test_superover_grid_layout():
    name, pots, amp = bench.grid_inputs("superover_grid", 0, 1, 8192, 16)
    assert name == "superover_var" and amp == 1.0 and pots.shape == (8192, 3)
    assert pots[:, 0].max() < 1.0                    # drive = 1.0 makes the variable-pot model singular
    assert len({tuple(p) for p in pots}) == 8192     # 32 x 16 x 16 distinct cells
    # the 4 instances of a wavefront (and the 16 of a block) differ only in the level pot
    blocks = pots.reshape(512, 16, 3)
    assert (blocks[:, :, :2] == blocks[:, :1, :2]).all()
    # weak scaling: rank r of W takes the r-th contiguous slice of the W-times larger grid
    _, p1, _ = bench.grid_inputs("superover_grid", 1, 2, 8192, 16)
    _, pall, _ = bench.grid_inputs("superover_grid", 0, 1, 16384, 16)
    assert np.array_equal(p1, pall[8192:])


def test_traffic_lookup_and_byte_model():
    r = bench.pmc_record("superover_grid", 8192, 44100)          # committed PMC pass of the headline
    assert r is not None and bench.pmc_traffic(r) > 1.4e10
    assert bench.pmc_record("superover_grid", 8192, 123) is None and bench.pmc_traffic(None) is None
    # the steady-state counters the bench line's roofline object quotes (tools/profile_gpu.sh, tools/merge_pmc.py)
    assert 0.0 < bench.pmc_lds_bank_conflict_frac(r) < 0.02
    assert 0.6 < bench.pmc_valu_issue_frac(r) < 1.0
    assert bench.pmc_lds_bank_conflict_frac(None) is None
    from helpers import load
    m = load("superover_var")
    assert bench.algorithmic_bytes(m, 8192, 44100) == 8192 * 44100 * 40 + 8192 * 2 * 8 * (11 + 11 + 13)


def test_pmc_record_never_quotes_a_stale_or_foreign_pass():
    """A bench line quotes PMC figures only from a pass of ITS configuration: same workload, size, solver stack and kernel
    variant, profiled launches within 3 % of its own kernel time (the cache-less line quoted the caching
    run's counters, the diode-clipper record was a round older than its kernel)."""
    r = bench.pmc_record("superover_grid", 8192, 44100)
    ms = r["kernel_avg_ms_profiled"]
    assert bench.pmc_record("superover_grid", 8192, 44100, kernel_ms=ms * 1.02) is r or \
        bench.pmc_record("superover_grid", 8192, 44100, kernel_ms=ms * 1.02) == r
    assert bench.pmc_record("superover_grid", 8192, 44100, kernel_ms=ms * 1.05) is None
    assert bench.pmc_record("superover_grid", 8192, 44100, kernel_ms=ms * 2.1) is None       # (the 624 ms cache-less run)
    assert bench.pmc_record("superover_grid", 8192, 44100, solver="no such stack") is None
    assert bench.pmc_record("superover_grid", 8192, 44100, kernel="acme_run_kernel<Shape<1,2,3,4,5,6>>") is None
    if r.get("solver"):      # records written from round 5 on carry what they ran
        assert bench.pmc_record("superover_grid", 8192, 44100, solver=r["solver"], kernel=r["kernel"], kernel_ms=ms) == r
    # executed fp64 work: 64 lanes x (2 FMA + MUL + ADD + TRANS)
    fake = dict(sq_insts_valu_fma_f64_per_launch=10.0, sq_insts_valu_mul_f64_per_launch=3.0, sq_insts_valu_add_f64_per_launch=2.0,
                sq_insts_valu_trans_f64_per_launch=1.0)
    assert bench.pmc_fp64_executed_flops(fake) == 64.0 * 26.0 and bench.pmc_fp64_executed_flops({}) is None


def test_montecarlo_models_are_seeded_per_rank():
    a = bench.montecarlo_models(0, 3)
    b = bench.montecarlo_models(0, 3)
    c = bench.montecarlo_models(1, 3)
    assert np.array_equal(a.d["a"], b.d["a"]) and not np.array_equal(a.d["a"], c.d["a"])
    assert (a.d["nns"], a.d["nqs"], a.d["nps"]) == ([7], [14], [5])


def test_cpu_baseline_handles_every_workload_shape():
    """cpu_baseline with per-instance pots (superover grid), per-instance amplitudes (diode clipper) and
    one common scalar amplitude (the Monte-Carlo workload, which once crashed here); the workloads without a fixture file (the
    clipper chains: their model is derived on the spot, also in the oracle's worker processes)."""
    import bench
    from helpers import HS, load
    for workload, n in (("superover_grid", 256), ("diodeclipper_sweep", 64), ("superover_montecarlo", 64), ("clipper_chain_20", 64), ("clipper_chain_34", 16)):
        fixture, pots, amp = bench.grid_inputs(workload, 0, 1, n, 64)
        model = bench.workload_model(workload, fixture, HS) if fixture.startswith("chain:") else load(fixture)
        rec = bench.cpu_baseline(fixture, model, pots, amp, 64, per_core=1)
        assert rec["value"] > 0 and rec["kind"] == "port" and rec["iters_per_sample"] >= 1.0


def test_drive_one_corner_is_singular():
    """Why bench.py's grid takes drive = i/32 instead of SURVEY 8(d)'s linspace(0, 1, 32): with the
    drive pot at exactly 1.0 its shorted leg (pins 2 and 3 of p1 share a node,
    examples/superover.jl:37) has 0 Ohm across a short -- the current through it is indeterminate
    and the variable-pot model's Jacobian is singular.  The oracle (the reference's algorithm)
    then fails to converge on practically every sample (one warning and ~900 Newton iterations per
    sample), with either solver stack; at drive = 31/32 and at drive = 0 it converges in a handful.
    (The reference's own test touches the corner for one sample only: the pot ramps of
    test/runtests.jl:778 start at 1.0.)"""
    from helpers import load, sine
    from oracle.refpy import RefRunner
    from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver
    T = 60
    for solver in (HomotopySolver, CachingHomotopySolver):
        m = load("superover_var", solver)
        for drive, singular in ((1.0, True), (31 / 32, False), (0.0, False)):
            u = np.zeros((4, T))
            u[0], u[1], u[2], u[3] = sine(T), drive, 0.5, 0.5
            r = RefRunner(m)
            r.run(u)
            if singular:
                assert r.report.n_warn >= T - 5 and r.report.iters_total > 300 * T
            else:
                assert r.report.n_warn == 0 and r.report.iters_total < 20 * T


def test_native_cpu_leg_times_the_native_build():
    """The '-O3 -march=native' leg of cpu_baseline must really load that build, also in a process that
    already holds the default -O2 library (it once silently re-timed the -O2 build)."""
    import os
    import bench
    from oracle import refpy
    os.environ.pop("ACME_REF_LIB", None)
    default = refpy.lib()
    native = bench.build_native_oracle()
    assert native is not None
    try:
        sig = np.zeros((1, 8))
        # _cpu_worker asserts that the library it is about to time is the one requested
        bench._cpu_worker(("diodeclipper", [sig], 8, None, False, native))
        assert os.environ.get("ACME_REF_LIB") == native
        assert refpy.lib().acme_path == os.path.realpath(native) and refpy.lib() is not default
        bench._cpu_worker(("diodeclipper", [sig], 8, None, False, None))
        assert refpy.lib() is default
    finally:
        os.environ.pop("ACME_REF_LIB", None)
        os.remove(native)
