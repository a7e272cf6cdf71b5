"""GPU-less logic tests: the *same kernel source* executed by the CPU wave emulator
(tests/emu) against the oracle.  These do not replace the -m gpu parity tests; they catch
logic errors (and cross-lane operations in divergent control flow) without a GPU."""
import numpy as np
import pytest

from helpers import assert_close, load, oracle_run, sweep_inputs


def emu_runner(emu_lib, model, n, **kw):
    from acme_jl_amd.runner import ModelRunner
    return ModelRunner(model, n, lib=emu_lib, **kw)


@pytest.mark.parametrize("name,N,T", [
    ("diodeclipper", 19, 200),
    ("superover_fixed", 5, 150),
    ("superover_var", 4, 100),
    ("birdie_fixed", 4, 200),
    ("birdie_var", 6, 200),
    ("rc_ladder", 3, 64),
])
def test_emulated_kernel_matches_oracle(emu_lib, name, N, T):
    m = load(name)
    u = sweep_inputs(name, N, T)
    r = emu_runner(emu_lib, m, N)
    y = r.run(u)
    yref, _ = oracle_run(m, u)
    assert_close(y, yref)


def test_emulated_state_roundtrip(emu_lib):
    m = load("birdie_fixed")
    u = sweep_inputs("birdie_fixed", 3, 120)
    y_once = emu_runner(emu_lib, m, 3).run(u)
    r = emu_runner(emu_lib, m, 3)
    y_split = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
    assert np.array_equal(y_once, y_split)
    x, p, z = r.get_state()
    assert x.shape == (3, m.nx) and p.shape == (3, 2) and z.shape == (3, 2)
