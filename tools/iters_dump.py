#!/usr/bin/env python
"""Dump the per-instance Newton iteration totals of the bench grid (second by second) to
gpurun_out/<tag>/iters.npy: the raw material for studying launch-time imbalance offline
(a launch lasts as long as its slowest SIMD).  usage (GPU box): python tools/iters_dump.py <tag> [seconds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

tag = sys.argv[1]
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, T = 8192, 44100
fixture, pots, amp = bench.grid_inputs("superover_grid", 0, 1, N, T)
m = DiscreteModel.load(os.path.join(ROOT, "tests", "golden", fixture + ".json"), solver=CachingHomotopySolver)
dev = torch.device("cuda", 0)
u = bench.make_u(torch, dev, m, pots, amp, N, T)
y = torch.empty((N, T, 1), dtype=torch.float64, device=dev)
r = ModelRunner(m, N, device=0)
out = []
for s in range(secs):
    r.reset_report()
    r.kernel_time(reset=True)
    r.run_torch(u, y)
    torch.cuda.synchronize()
    ms, _ = r.kernel_time()
    it = r.report_arrays()["iters_total"].copy()
    out.append(it)
    w = it.reshape(-1, 4).max(axis=1)
    print(f"second {s + 1}: kernel {ms:.1f} ms, its/sample mean {it.mean() / T:.3f}, per-wave(max of 4) mean {w.mean() / T:.3f} "
          f"max {w.max() / T:.3f}")
os.makedirs(os.path.join(ROOT, "gpurun_out", tag), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", tag, "iters.npy"), np.array(out))
