#!/usr/bin/env python
"""developer: where do the literal path's two image placements (IMGL = 1: staged in LDS, 0: read from HBM) of the 27-unknown
chain part?  First differing output sample per instance, its size, and the iteration counts around it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from helpers import mid_size_models
    from acme_jl_amd.model import CachingHomotopySolver
    from acme_jl_amd.runner import ModelRunner
    name, m, u5 = [c for c in mid_size_models(more=True) if c[0].startswith("27")][0]
    N, T = 70, u5.shape[2]
    u = np.logspace(-1.5, 0.6, N)[:, None, None] * u5[2:3] / np.abs(u5[2]).max()
    m.solver = CachingHomotopySolver
    r = ModelRunner(m, N)
    ys, its = [], []
    done = np.zeros(N, dtype=np.int64)
    for n in range(T):
        print("sample", n, flush=True)
        ys.append(r.run(u[:, :, n:n + 1]))
        tot = r.report_arrays()["iters_total"].copy()
        its.append(tot - done)
        done = tot
    np.save(sys.argv[2], np.concatenate(ys, axis=2))
    np.save(sys.argv[2] + ".its", np.stack(its, axis=1))
    sys.exit(0)
for lit, imgl in (("0", "1"), ("0", "0"), ("1", "1"), ("1", "0")):
    env = dict(os.environ, ACME_COOP_LITERAL=lit, ACME_COOP_REG="1", ACME_COOP_WPB="1", ACME_COOP_GPW="4", ACME_COOP_IMGL=imgl)
    f = f"/tmp/lit{lit}_imgl{imgl}"
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", f], env=env, timeout=900, capture_output=True, text=True)
    last = [l for l in p.stdout.split("\n") if l.startswith("sample")][-1:]
    print("literal" if lit == "1" else "threshold", "IMGL", imgl, "exit", p.returncode, "last", last, [l for l in p.stderr.split("\n") if "fault" in l][:1], flush=True)
