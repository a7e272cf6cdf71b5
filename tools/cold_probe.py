#!/usr/bin/env python
"""developer: the first second of a FRESH headline batch as one launch and as two (a short first launch gives the second
one measured wave costs to place by): is the cold step's excess the missing placement or the empty solution caches?
usage (GPU box): python tools/cold_probe.py [first-launch samples ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel
from acme_jl_amd.runner import ModelRunner
n, T = 8192, 44100
dev = torch.device("cuda", 0)
fixture, pots, amp = bench.grid_inputs("superover_grid", 0, 1, n, T)
model = DiscreteModel.load(os.path.join(ROOT, "tests", "golden", fixture + ".json"), solver=CachingHomotopySolver)
u = bench.make_u(torch, dev, model, pots, amp, n, T)
y = torch.empty((n, T, model.ny), dtype=torch.float64, device=dev)
for split in [0] + [int(a) for a in sys.argv[1:]] + [0]:
    r = ModelRunner(model, n, device=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if split:
        ya = r.run_torch(u[:, :split].contiguous())
        yb = r.run_torch(u[:, split:].contiguous())
    else:
        r.run_torch(u, y)
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    ms, launches = r.kernel_time()
    t0 = time.perf_counter()
    r.run_torch(u, y)
    torch.cuda.synchronize()
    print(f"first launch {split or T} samples: cold second {1e3 * cold:.1f} ms wall, kernels {ms:.1f} ms in {launches} launches; next step {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
