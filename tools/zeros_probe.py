#!/usr/bin/env python
"""developer: the any-size mid-size kernel of commit ad793d5 built WITHOUT its register bound computed
zeros whenever a wave carried fewer than four instances.  Runs every build_variants/libacme_hip_z*.so (the failing build and
its bisection variants) on the 24-unknown clipper chain with 4, 2 and 1 instances per wave and prints what came out.
usage (GPU box): python tools/zeros_probe.py            (re-invokes itself once per library: ACME_HIP_LIB is read at import)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] in ("--one", "--private"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from helpers import HS, mid_size_models
    from acme_jl_amd.runner import ModelRunner
    name, m, u5 = mid_size_models()[0]
    N, T = 70, u5.shape[2]
    u = np.logspace(-1.5, 0.6, N)[:, None, None] * u5[2:3] / np.abs(u5[2]).max()
    m.solver = HS
    os.environ["ACME_COOP_REG"] = "0"
    os.environ["ACME_COOP_WPB"] = "1"
    private = sys.argv[1] == "--private"          # (one model image per instance: the image then stays in HBM at every group count)
    for gpw in ("4", "2", "1"):
        os.environ["ACME_COOP_GPW"] = gpw
        r = ModelRunner(m, N, models=[m] * N if private else None)
        y = r.run(u)
        ra = r.report_arrays()
        x, p, z = r.get_state()
        print(f"  gpw {gpw}: sum|y| {np.abs(y).sum():.12e}  zeros {int((y == 0).sum())}/{y.size}  nan {int(np.isnan(y).sum())}  "
              f"iters {int(ra['iters_total'].sum())}  warn {int(ra['n_warn'].sum())}  sum|x| {np.abs(x).sum():.6e}  sum|z| {np.abs(z).sum():.6e}  "
              f"per-instance zero rows {[int(i) for i in np.where((y == 0).all(axis=(1, 2)))[0][:12]]}", flush=True)
        if "diag" in os.environ.get("ACME_HIP_LIB", ""):
            for inst in (0, 1, 5):
                print(f"    diag inst {inst}: sum|img in LDS| {y[inst,0,0]:.6e} sum|img in HBM| {y[inst,0,1]:.6e} sum|tables in LDS| {y[inst,0,2]:.6e} "
                      f"W - lds {y[inst,0,3]:.0f} shared doubles {y[inst,0,4]:.0f} last u in LDS {y[inst,0,5]:.6e} last u in HBM {y[inst,0,6]:.6e} O.total {y[inst,0,7]:.0f}", flush=True)
    sys.exit(0)
libs = sorted(glob.glob(os.path.join(ROOT, "build_variants", "libacme_hip_z*.so")))
for lib in libs:
    print("==", os.path.basename(lib), flush=True)
    env = dict(os.environ, ACME_HIP_LIB=lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, timeout=600)
