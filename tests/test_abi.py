"""The C-ABI library loads on a GPU-less box and exports every symbol include/acme_hip.h
declares (no compute calls here)."""
import os
import re

from helpers import ROOT


def test_header_symbols_are_exported():
    from acme_jl_amd import runner
    hdr = open(os.path.join(ROOT, "include", "acme_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(acme_[a-z_0-9]+)\s*\(", hdr)))
    assert set(declared) == set(runner.ABI_SYMBOLS), set(declared) ^ set(runner.ABI_SYMBOLS)
    lib = runner.Library(runner.DEFAULT_LIBRARY)
    for sym in declared:
        assert hasattr(lib.L, sym), sym


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, never fall back."""
    import pytest
    from acme_jl_amd import runner
    from helpers import load
    lib = runner.Library(runner.DEFAULT_LIBRARY)
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(runner.AcmeError, match="no HIP device"):
        runner.ModelRunner(load("diodeclipper"), 1, lib=lib)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "acme_jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle/acme_ref", "").lower() or f == "hostsolve.py" or \
                    "import oracle" not in text and "from oracle" not in text, f
                assert "import oracle" not in text and "from oracle" not in text and "libacme_emu" not in text, f


def test_plain_c_client_links_and_fails_loudly_without_gpu():
    """examples/abi_demo.c (plain C over include/acme_hip.h) builds against the product library;
    on a GPU-less box it must stop with the no-device error, not compute anything."""
    import subprocess
    import pytest
    import __graft_entry__ as g
    from acme_jl_amd import runner
    exe = g.build_abi_demo()
    if runner.Library(runner.DEFAULT_LIBRARY).device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1
    assert "no HIP device" in out.stderr and "no CPU fallback" in out.stderr
