#!/usr/bin/env python
"""developer: iteration totals of the mid-size kernel's literal path on the 27-unknown chain (caching stack) per launch shape,
against the oracle's -- a few repetitions each (is a difference reproducible?).  GPU box; ACME_HIP_LIB selects a build."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import mid_size_models, oracle_run
from acme_jl_amd.model import CachingHomotopySolver
from acme_jl_amd.runner import ModelRunner
name, m, u5 = [c for c in mid_size_models(more=True) if c[0].startswith(sys.argv[1] if len(sys.argv) > 1 else "27")][0]
N, T = 70, u5.shape[2]
u = np.logspace(-1.5, 0.6, N)[:, None, None] * u5[2:3] / np.abs(u5[2]).max()
m.solver = CachingHomotopySolver
yref, its = oracle_run(m, u, cache_limit=16)
pins = ("ACME_COOP_REG", "ACME_COOP_WPB", "ACME_COOP_GPW", "ACME_COOP_IMGL")
for lit in ("1", "0"):
    os.environ["ACME_COOP_LITERAL"] = lit
    for reg in ("1", "0"):
        for shape in ("", "110", "120", "140", "411"):
            for k in pins: os.environ.pop(k, None)
            for k, v in zip(pins, (reg,) + tuple(shape)): os.environ[k] = v
            out = []
            for rep in range(3):
                r = ModelRunner(m, N)
                y = np.concatenate([r.run(u[:, :, :50]), r.run(u[:, :, 50:])], axis=2)
                it = r.report_arrays()["iters_total"]
                d = np.where(it != its)[0]
                out.append((float(np.abs(y - yref).max()), [(int(i), int(it[i] - its[i])) for i in d[:6]]))
            print(name, "literal" if lit == "1" else "threshold", "reg", reg, "shape", shape or "auto", out, flush=True)
