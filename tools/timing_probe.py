#!/usr/bin/env python3
"""In-situ time breakdown of the run kernel (developer tool, GPU box only).

Needs a library built with -DACME_TIMING (tools/variants.sh build "timing:-DACME_TIMING") and
ACME_HIP_LIB pointing at it: that build accumulates shader-clock cycles per code region in every
wave and writes the totals over the first samples of y.  Prints cycles per sample and per Newton
iteration for each region, averaged over the waves of the superover grid."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from acme_jl_amd import runner as R  # noqa: E402

NAMES = ["post(y,x,io)", "pre(p)", "setup(set_p,extrap)", "evaluate", "pivot+adopt", "GJ<0>", "GJ<NP>",
         "origin store", "newton glue", "homotopy glue", "  eval: q=pf+fq*z", "  eval: exp x2", "  eval: element rows"]


def main(T=2205, n=8192):
    dev = torch.device("cuda:0")
    workload = os.environ.get("ACME_PROBE_WORKLOAD", "superover_grid")      # or birdie_grid, diodeclipper_sweep
    fixture, pots, amp = bench.grid_inputs(workload, 0, 1, n, T)
    from acme_jl_amd.model import DiscreteModel
    from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver
    solver = HomotopySolver if os.environ.get("ACME_PROBE_SOLVER") == "homotopy" else CachingHomotopySolver
    model = DiscreteModel.load(os.path.join(bench.ROOT, "tests", "golden", fixture + ".json"), solver=solver)
    u = bench.make_u(torch, dev, model, pots, amp, n, T, 176400 if workload == "birdie_grid" else bench.FS)
    r = R.ModelRunner(model, n)
    y = r.run_torch(u)
    torch.cuda.synchronize()
    iters = float(np.sum(r.report_arrays()["iters_total"])) / n
    tb = y[:, :len(NAMES), 0].double().cpu().numpy()      # [n][bucket]
    per_wave = tb[::4]                                       # 4 instances per wave hold the same numbers
    tot = per_wave.sum(axis=1).mean()
    print(f"T={T} instances={n} iterations/sample={iters / T:.3f} kernel_ms={r.last_kernel_ms():.2f}")
    print(f"{'region':24s} {'cycles/sample':>14s} {'cycles/iter':>12s} {'share':>7s}")
    for i, name in enumerate(NAMES):
        c = per_wave[:, i].mean()
        print(f"{name:24s} {c / T:14.1f} {c / iters:12.1f} {100 * c / tot:6.1f}%")
    print(f"{'total':24s} {tot / T:14.1f} {tot / iters:12.1f}")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
