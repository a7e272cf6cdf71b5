#!/usr/bin/env python
"""The pathological-instance tail (SURVEY 7 "data-dependent trip counts"), measured.  GPU box only.

SURVEY 8(d)'s literal config-3 grid has drive in linspace(0, 1, 32): its last column -- 256 of the 8 192 instances --
sits on the singular drive = 1.0 corner of the variable-pot superover (the pot's shorted leg has 0 Ohm: the current
through it is indeterminate), where the reference's solver stack fails on practically every sample after ~900 Newton
iterations (tests/test_bench_helpers.py::test_drive_one_corner_is_singular).  A launch lasts as long as its slowest
wave, and every wave is one serial recurrence over the samples.  This probe runs T samples of

  * the bench's grid (drive = i / 32: no singular cell),
  * the literal grid (drive = i / 31),
  * the literal grid's 7 936 healthy instances alone,
  * the literal grid with the healthy instances isolated from the slow ones (acme_batch_set_isolation), if built,

and prints launch times, iteration statistics and warning counts.
usage: python tools/tail_probe.py [samples]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acme_jl_amd.model import CachingHomotopySolver, DiscreteModel  # noqa: E402
from acme_jl_amd.runner import ModelRunner  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 220        # (a singular cell costs ~50 ms of GPU time per sample)
N = 8192
dev = torch.device("cuda", 0)
model = DiscreteModel.load(os.path.join(ROOT, "tests", "golden", "superover_var.json"), solver=CachingHomotopySolver)
idx = np.arange(N)
level, tone = (idx % 16) / 15.0, ((idx // 16) % 16) / 15.0
sig = np.sin(2 * np.pi * 1000 / 44100 * np.arange(T))


def inputs(drive, sel=None):
    pots = np.stack([drive, tone, level], axis=1)
    if sel is not None:
        pots = pots[sel]
    u = torch.zeros((len(pots), T, 4), dtype=torch.float64, device=dev)
    u[:, :, 0] = torch.from_numpy(sig).to(dev)[None, :]
    u[:, :, 1:] = torch.from_numpy(pots).to(dev)[:, None, :]
    return u


def run(name, u, isolate=None):
    import warnings
    r = ModelRunner(model, u.shape[0], device=0)
    if isolate is not None:
        r.set_isolation(isolate)
    ys = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rep in range(2):          # second call: the classification of the first one's iteration counts applies
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = r.run_torch(u)
            torch.cuda.current_stream().synchronize()       # the caller's stream: the fast group with isolation on
            t_fast = time.perf_counter() - t0
            r.wait(check=False)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            ys.append(y.clone())
            ra = r.report_arrays()
            its = np.asarray(ra["iters_total"], dtype=float) / (T * (rep + 1))
            print(f"{name:44s} call {rep + 1}: healthy results after {1e3 * t_fast:9.1f} ms, everything after {1e3 * t_all:9.1f} ms; "
                  f"iterations/sample median {np.median(its):.2f} max {its.max():.1f}; instances with warnings {int((np.asarray(ra['n_warn']) > 0).sum())}, "
                  f"warnings {int(np.asarray(ra['n_warn']).sum())}", flush=True)
    return ys


bench_grid = run("bench grid (drive = i/32)", inputs((idx // 256) / 32.0))
drive_lit = (idx // 256) / 31.0
healthy = drive_lit < 1.0
y_lit = run("literal grid (drive = i/31, 256 singular)", inputs(drive_lit))
y_h = run("literal grid, the 7 936 healthy ones alone", inputs(drive_lit, healthy))
for a, b in zip(y_lit, y_h):
    assert torch.equal(a[torch.from_numpy(healthy).to(dev)], b), "healthy instances must not depend on their neighbours"
print("healthy instances: bit-identical with and without the singular column in the batch")
if hasattr(ModelRunner, "set_isolation"):
    y_iso = run("literal grid, isolation at 50 its/sample", inputs(drive_lit), isolate=50.0)
    for a, b in zip(y_lit, y_iso):
        assert torch.equal(a, b), "isolation must not change any result"
    print("isolation: bit-identical results")
