#!/usr/bin/env python
"""Throughput of the generic (padded, all-element-kinds) kernel shapes, which any circuit that is not
one of the BASELINE models runs in (never measured).  GPU box only.
usage: python tools/generic_shape_probe.py [instances] [samples]"""
import os
import sys
import time
from fractions import Fraction

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4410
t = Fraction(1, 44100)
cases = [
    ("clipper chain, 2 stages (nn 4)", DiscreteModel(circuits.clipper_chain(2), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    ("clipper chain, 4 stages (nn 8)", DiscreteModel(circuits.clipper_chain(4), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    ("clipper chain, 8 stages (nn 16)", DiscreteModel(circuits.clipper_chain(8), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    # (stages separated by op-amp buffers: one sub-problem per stage -- the directly coupled chain above does not decompose)
    ("buffered clipper chain, 4 stages (nsub 4)", DiscreteModel(circuits.buffered_clipper_chain(4), t, CachingHomotopySolver)),
    # beyond the tuned shapes: the generic lane-per-instance kernel (acme_generic.h)
    ("clipper chain, 10 stages (nn 20) [generic]", DiscreteModel(circuits.clipper_chain(10), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    # (up to 8 sub-problems: a tuned shape since round 6; 9 and more: the lane-per-instance generic kernel)
    ("buffered clipper chain, 6 stages (nsub 6)", DiscreteModel(circuits.buffered_clipper_chain(6), t, CachingHomotopySolver)),
    ("buffered clipper chain, 8 stages (nsub 8)", DiscreteModel(circuits.buffered_clipper_chain(8), t, CachingHomotopySolver)),
    ("buffered clipper chain, 9 stages (nsub 9) [generic]", DiscreteModel(circuits.buffered_clipper_chain(9), t, CachingHomotopySolver)),
    # the cooperative mid-size kernel's range (acme_coop.h; ACME_COOP=0: the lane-per-instance kernel)
    ("clipper chain, 12 stages (nn 24) [generic]", DiscreteModel(circuits.clipper_chain(12), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    ("clipper chain, 16 stages (nn 32) [generic]", DiscreteModel(circuits.clipper_chain(16), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    # beyond the register instantiations: the any-size instantiation
    ("clipper chain, 17 stages (nn 34) [generic]", DiscreteModel(circuits.clipper_chain(17), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    ("clipper chain, 24 stages (nn 48) [generic]", DiscreteModel(circuits.clipper_chain(24), t, CachingHomotopySolver, decompose_nonlinearity=False)),
    ("clipper chain, 32 stages (nn 64) [generic]", DiscreteModel(circuits.clipper_chain(32), t, CachingHomotopySolver, decompose_nonlinearity=False)),
]
if len(sys.argv) > 3:      # a subset by substring (underscores stand for blanks)
    sys.argv[3] = sys.argv[3].replace("_", " ")
    cases = [c for c in cases if sys.argv[3] in c[0]]
dev = torch.device("cuda", 0)
sig = torch.sin(2 * np.pi * 1000 / 44100 * torch.arange(T, dtype=torch.float64, device=dev))
amp = torch.logspace(-2, 0.7, N, dtype=torch.float64, device=dev)
for name, m in cases:
    try:
        r = ModelRunner(m, N, device=0)
    except Exception as e:      # e.g. LDS limit of the caching solver on the largest shape
        m.solver = "HomotopySolver{SimpleSolver}"
        r = ModelRunner(m, N, device=0)
        name += " [no cache: " + str(e)[:40] + "...]"
    u = torch.zeros((N, T, m.nu), dtype=torch.float64, device=dev)
    u[:, :, 0] = amp[:, None] * sig[None, :] * (0.3 if m.nu > 1 else 1.0)
    if m.nu > 1:
        u[:, :, 1] = 0.5
    y = r.run_torch(u)
    torch.cuda.synchronize()
    r.reset_report()
    r.kernel_time(reset=True)
    t0 = time.perf_counter()
    r.run_torch(u, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ra = r.report_arrays()
    print(f"{name:48s} shape {r.kernel_shape()} {r.kernel_family()} [{len(m.subs)} sub]: {N * T / dt:.3e} inst*samples/s, "
          f"{ra['iters_total'].sum() / (N * T):.2f} its/sample, warnings {int(ra['n_warn'].sum())}, finite {bool(torch.isfinite(y).all())}",
          flush=True)
