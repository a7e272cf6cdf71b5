"""Regenerate the model-block fixtures tests/golden/*.json from the host front end.

These are outputs of THIS repository's exact-rational derivation (acme_jl_amd.derive), kept
as fixtures so that GPU-side tests and bench.py need not re-derive; tests/test_frontend.py
re-derives them on CPU and checks they are reproduced exactly.  The fixtures name the cache-less
HomotopySolver{SimpleSolver} stack (what the same-Newton-path parity tests compare at 1e-12); tests and
bench.py that want the reference's default caching stack pass it to DiscreteModel.load.  Run from the repo root:
    python tests/golden/make_models.py
"""
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acme_jl_amd import examples  # noqa: E402
from acme_jl_amd.model import DiscreteModel, HomotopySolver  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FS = Fraction(1, 44100)
MODELS = {
    "diodeclipper": lambda: DiscreteModel(examples.diodeclipper(), FS, HomotopySolver),
    "superover_fixed": lambda: DiscreteModel(examples.superover(1.0, 1.0, 1.0), FS, HomotopySolver),
    "superover_var": lambda: DiscreteModel(examples.superover(), FS, HomotopySolver),
    "birdie_fixed": lambda: DiscreteModel(examples.birdie(vol=0.8), FS, HomotopySolver),
    "birdie_var": lambda: DiscreteModel(examples.birdie(), FS, HomotopySolver),
    "birdie_var_176k": lambda: DiscreteModel(examples.birdie(), Fraction(1, 176400), HomotopySolver),
    "rc_ladder": lambda: DiscreteModel(examples.rc_ladder(), FS, HomotopySolver),
    "sallenkey": lambda: DiscreteModel(examples.sallenkey(), FS, HomotopySolver),
}

if __name__ == "__main__":
    for name, make in MODELS.items():
        m = make().tune_row_order()
        m.save(os.path.join(HERE, name + ".json"))
        print(name, "nx", m.nx, "nu", m.nu, "ny", m.ny, [(s.nn, s.nq, s.np) for s in m.subs])
