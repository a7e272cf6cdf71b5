#!/usr/bin/env python
"""Cycles per code region of the cooperative mid-size kernel (acme_coop.h), per wave and sample: needs a library built with
-DACME_COOP_TIMING (ACME_HIP_LIB).   usage (GPU box): ACME_HIP_LIB=... python tools/coop_timing_probe.py [stages] [instances] [samples]"""
import os
import sys
from fractions import Fraction

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

stages = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1102
NAMES = ["set_p", "extrapolation (solve!)", "evaluate!", "setlhs! (LU)", "solve! + step", "accept (Jp, origin)", "cache lookup",
         "y / x update", "p = dq x + eq u", "rest", "(LU: pivot search | LDS matrix: rest of the steps)",
         "(LU: pivot row hand-off, 1 / pivot | LDS matrix: factorisations [count])", "LDS step: loads + scan (one instance per wave: requesting the loads)", "LDS step: 1 / pivot, column k (...: waiting for the scan)",
         "LDS step: pairs read ahead (...: 1 / pivot, multipliers)", "LDS step: other pairs (...: stores)", "LDS step: chunks [count] (...: unforeseen pairs)", "LDS steps [count]"]
m = DiscreteModel(circuits.clipper_chain(stages), Fraction(1, 44100), CachingHomotopySolver, decompose_nonlinearity=False)
dev = torch.device("cuda", 0)
sig = torch.sin(2 * np.pi * 1000 / 44100 * torch.arange(T, dtype=torch.float64, device=dev))
amp = torch.logspace(-2, 0.7, N, dtype=torch.float64, device=dev)
r = ModelRunner(m, N, device=0)
u = torch.zeros((N, T, m.nu), dtype=torch.float64, device=dev)
u[:, :, 0] = amp[:, None] * sig[None, :]
y = r.run_torch(u)
torch.cuda.synchronize()
its = float(r.report_arrays()["iters_total"].sum()) / (N * T)
tb = y[:, :len(NAMES), 0].cpu().numpy()
tot = tb.sum(axis=1).mean()
print(f"{r.kernel_family()} kernel, nn = {2 * stages}: {N} instances x {T} samples, {its:.2f} iterations per sample, kernel {r.last_kernel_ms():.1f} ms")
for k, name in enumerate(NAMES):
    cyc = tb[:, k].mean()
    print(f"{name:26s} {cyc / T:10.0f} cycles per sample  {100 * cyc / tot:5.1f} %")
print(f"{'total':26s} {tot / T:10.0f}")
