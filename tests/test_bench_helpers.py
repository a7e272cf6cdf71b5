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
    assert bench.pmc_traffic("superover_grid", 8192, 44100) > 1.4e10     # committed PMC pass
    assert bench.pmc_traffic("superover_grid", 8192, 123) is None
    from helpers import load
    m = load("superover_var")
    assert bench.algorithmic_bytes(m, 8192, 44100) == 8192 * 44100 * 40 + 8192 * 2 * 8 * (11 + 11 + 13)


def test_montecarlo_models_are_seeded_per_rank():
    a = bench.montecarlo_models(0, 3)
    b = bench.montecarlo_models(0, 3)
    c = bench.montecarlo_models(1, 3)
    assert np.array_equal(a.d["a"], b.d["a"]) and not np.array_equal(a.d["a"], c.d["a"])
    assert (a.d["nns"], a.d["nqs"], a.d["nps"]) == ([7], [14], [5])


def test_cpu_baseline_handles_every_workload_shape():
    """cpu_baseline with per-instance pots (superover grid), per-instance amplitudes (diode clipper) and
    one common scalar amplitude (the Monte-Carlo workload, which once crashed here)."""
    import bench
    from helpers import load
    for workload, n in (("superover_grid", 256), ("diodeclipper_sweep", 64), ("superover_montecarlo", 64)):
        fixture, pots, amp = bench.grid_inputs(workload, 0, 1, n, 64)
        rec = bench.cpu_baseline(fixture, load(fixture), pots, amp, 64, per_core=1)
        assert rec["value"] > 0 and rec["kind"] == "port" and rec["iters_per_sample"] >= 1.0
