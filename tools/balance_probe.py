#!/usr/bin/env python3
"""How much of a launch of the headline grid is load imbalance between SIMDs (developer tool, GPU box).

A launch ends with its slowest wave; every wave carries its 4 instances through the whole step, two waves share a SIMD.
The probe measures the per-wave work of a steady-state step (Newton iterations), then re-runs the same grid with the
WAVES dealt to the launch's slots in different orders -- the instances are identified by their rows of u alone, so a
permutation of u is a permutation of the waves -- and times the same steady-state steps:
  natural      the bench's order (drive slowest, level fastest)
  sorted       heaviest waves first
  pair+1024    heaviest with lightest on slots q and q + 1024 (wave i of blocks b and b + 256)
  pair+4       ... on slots q and q + 4 (wave i of blocks 2c and 2c + 1)
  pair+1       ... on neighbouring waves of a block
  random       a random deal
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

dev = torch.device("cuda:0")
N, T = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 22050
WARM, TIMED = 5, 2
fixture, pots, amp = bench.grid_inputs("superover_grid", 0, 1, N, T)
m = DiscreteModel.load(os.path.join(bench.ROOT, "tests", "golden", fixture + ".json"), solver=CachingHomotopySolver)
NW = N // 4


def run(perm_waves):
    """steady-state ms per step, per-wave iterations of the last step (in the order of perm_waves' slots)"""
    inst = (perm_waves[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
    r = ModelRunner(m, N)
    ms_all, prev, d = [], np.zeros(N), None
    for s in range(WARM + TIMED):
        t = torch.arange(s * T, (s + 1) * T, dtype=torch.float64, device=dev)
        u = torch.empty((N, T, m.nu), dtype=torch.float64, device=dev)
        u[:, :, 0] = amp * torch.sin(2 * np.pi * 1000.0 / bench.FS * t)[None, :]
        u[:, :, 1:] = torch.as_tensor(pots[inst], dtype=torch.float64, device=dev)[:, None, :]
        r.kernel_time(reset=True)
        y = r.run_torch(u)
        torch.cuda.synchronize()
        ms, _ = r.kernel_time()
        it = r.report_arrays()["iters_total"].astype(float)
        d, prev = it - prev, it
        if s >= WARM:
            ms_all.append(ms)
        del u, y
    return float(np.mean(ms_all)), d.reshape(-1, 4)


nat = np.arange(NW)
ms0, d0 = run(nat)
w = d0.max(axis=1) / T                      # passes per sample of each wave (at least)
print(f"natural: {ms0:.2f} ms per {T} samples; wave passes/sample mean {w.mean():.3f} max {w.max():.3f} (max/mean {w.max() / w.mean():.3f});"
      f" instance iterations/sample mean {d0.mean() / T:.3f}")
pair = (w[:NW // 2 * 2].reshape(-1)).copy()
for name, stride in (("q,q+1024", 1024), ("q,q+4", 4), ("q,q+1", 1)):
    q = np.arange(NW)
    a = q[(q // stride) % 2 == 0]
    s = w[a] + w[a + stride]
    print(f"   natural order, SIMD load if waves {name} share a SIMD: mean {s.mean():.3f} max {s.max():.3f} (max/mean {s.max() / s.mean():.3f})")
order = np.argsort(-w)                      # heaviest first


def paired(stride):
    q = np.arange(NW)
    a = q[(q // stride) % 2 == 0]
    perm = np.empty(NW, dtype=np.int64)
    perm[a] = order[:NW // 2]
    perm[a + stride] = order[::-1][:NW // 2]
    return perm


rng = np.random.default_rng(7)
for name, perm in (("sorted", order), ("pair+1024", paired(1024)), ("pair+4", paired(4)), ("pair+1", paired(1)),
                   ("random", rng.permutation(NW))):
    ms, d = run(perm)
    print(f"{name:10s}: {ms:.2f} ms ({ms0 / ms:.3f} x natural); iterations/sample mean {d.mean() / T:.3f}")
