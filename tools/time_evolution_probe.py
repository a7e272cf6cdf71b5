#!/usr/bin/env python3
"""Run time and Newton iterations of consecutive 0.1 s launches of the bench grid (developer tool, GPU box):
how the cost of the workload itself evolves along the signal (mean and slowest-wave iterations)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

dev = torch.device("cuda:0")
N, T, STEPS = 8192, 4410, int(os.environ.get("PROBE_STEPS", "40"))
fixture, pots, amp = bench.grid_inputs("superover_grid", 0, 1, N, T * STEPS)
m = DiscreteModel.load(os.path.join(bench.ROOT, "tests", "golden", fixture + ".json"), solver=CachingHomotopySolver)
u = bench.make_u(torch, dev, m, pots, amp, N, T * STEPS)
r = ModelRunner(m, N)
prev = np.zeros(N)
for s in range(STEPS):
    r.kernel_time(reset=True)
    r.run_torch(u[:, s * T:(s + 1) * T].contiguous())
    torch.cuda.synchronize()
    ms, _ = r.kernel_time()
    it = r.report_arrays()["iters_total"].astype(float)
    d = it - prev
    prev = it
    w = d.reshape(-1, 4).max(axis=1)
    print(f"t={s * 0.1:4.1f}s kernel {ms:7.1f} ms  iters/sample mean {d.mean() / T:5.2f}  slowest wave {w.max() / T:5.2f}  max/mean {w.max() / w.mean():.3f}")
# steady state: which cells are the slow ones
w2 = d.reshape(-1, 4).max(axis=1) / T
order = np.argsort(-w2)[:12]
for k in order:
    i = k * 4
    print(f"wave {k:5d} drive {pots[i, 0]:.3f} tone {pots[i, 1]:.3f} levels {pots[i, 2]:.2f}..{pots[i + 3, 2]:.2f}: {w2[k]:.2f} iters/sample")
print("percentiles of iters/sample per wave (50,75,90,95,99,100):", np.percentile(w2, [50, 75, 90, 95, 99, 100]).round(2))
