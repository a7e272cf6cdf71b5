"""Long-run randomized parity sweep on the GPU (developer script, run through gpurun):
48 random superover instances x 44 100 samples against the oracle, for both solver stacks.
Lives under tests/ because it uses the oracle.   python tests/parity_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import load, oracle_run, assert_close
from acme_jl_amd.model import CachingHomotopySolver, HomotopySolver
from acme_jl_amd.runner import ModelRunner
rng=np.random.default_rng(123)
N,T=48,44100
s=np.sin(2*np.pi*1000/44100*np.arange(T))
for solver,lim in ((HomotopySolver,None),(CachingHomotopySolver,16)):
    m=load("superover_var", solver)
    u=np.zeros((N,4,T)); u[:,0]=rng.uniform(0.05,1.5,N)[:,None]*s; u[:,1]=rng.uniform(0,0.99,N)[:,None]; u[:,2]=rng.uniform(0,1,N)[:,None]; u[:,3]=rng.uniform(0,1,N)[:,None]
    r=ModelRunner(m,N); t0=time.time(); y=r.run(u); ra=r.report_arrays()
    yref,its=oracle_run(m,u,cache_limit=lim)
    err=np.abs(y-yref).max(axis=(1,2))
    print(solver, "max err", err.max(), "n_warn", ra["n_warn"].sum(), "iters gpu/oracle", ra["iters_total"].sum(), its.sum(), "worst instance rel iters diff", np.abs(ra["iters_total"]-its).max()/its.max())

# the other BASELINE circuits, one second each, both solver stacks
from helpers import sweep_inputs
for name, N, T in (("diodeclipper", 64, 44100), ("birdie_var_176k", 32, 176400), ("superover_fixed", 32, 44100)):
    for solver, lim in ((HomotopySolver, None), (CachingHomotopySolver, 16)):
        m = load(name, solver)
        u = sweep_inputs(name, N, T)
        r = ModelRunner(m, N)
        y = r.run(u)
        ra = r.report_arrays()
        yref, its = oracle_run(m, u, cache_limit=lim)
        print(name, solver, "max err", np.abs(y - yref).max(), "n_warn", ra["n_warn"].sum(), "iters gpu/oracle", ra["iters_total"].sum(), its.sum())
