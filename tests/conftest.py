import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library on a GPU box.  Fails loudly (never skips) if it cannot run:
    a GPU test that silently passed on a fallback would be worthless."""
    from acme_jl_amd import runner
    lib = runner.default_library()
    assert lib.device_count() > 0, "no HIP device visible to libacme_hip.so"
    return lib


@pytest.fixture(scope="session")
def emu_lib():
    """CPU wave-emulator build of the kernel source (tests/emu), for GPU-less logic tests."""
    import subprocess
    from acme_jl_amd import runner
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    return runner.Library(os.path.join(ROOT, "tests", "emu", "libacme_emu.so"))
