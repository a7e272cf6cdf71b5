#!/usr/bin/env python
"""developer: iteration totals of the cooperative kernel's instantiations on the mid-size test models (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import HS, mid_size_models, beyond_the_tuned_shapes
from acme_jl_amd.runner import ModelRunner
for name, m, u5 in mid_size_models() + beyond_the_tuned_shapes()[:1]:
    N, T = 70, u5.shape[2]
    u = np.logspace(-1.5, 0.6, N)[:, None, None] * u5[2:3] / np.abs(u5[2]).max()
    m.solver = HS
    out = {}
    for reg in ("1", "0"):
        for gpw in ("", "4", "2", "1"):
            os.environ["ACME_COOP_REG"] = reg
            if gpw: os.environ["ACME_COOP_GPW"] = gpw
            else: os.environ.pop("ACME_COOP_GPW", None)
            try:
                r = ModelRunner(m, N)
                y = r.run(u)
                it = r.report_arrays()["iters_total"]
                out[(reg, gpw)] = y
                print(name, "reg", reg, "gpw", gpw or "auto", "iters", int(it.sum()), "ysum %.12e" % float(np.abs(y).sum()), flush=True)
            except Exception as e:
                print(name, reg, gpw, "failed", str(e)[:80])
