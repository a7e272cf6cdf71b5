import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from acme_jl_amd.model import DiscreteModel, CachingHomotopySolver
from acme_jl_amd.runner import ModelRunner
dev=torch.device("cuda:0")
N,T=8192,4410
fixture,pots,amp=bench.grid_inputs("superover_grid",0,1,N,T)
m=DiscreteModel.load(os.path.join("tests","golden",fixture+".json"), solver=CachingHomotopySolver)
u=bench.make_u(torch,dev,m,pots,amp,N,T)
r=ModelRunner(m,N)
for rep in range(2):
    r.reset_report(); y=r.run_torch(u); torch.cuda.synchronize()
it=r.report_arrays()["iters_total"].astype(float)
w=it.reshape(-1,4).max(axis=1)      # per wave: lockstep -> max of its 4 instances (approx: sum of per-sample max unknown)
b=w.reshape(-1,4)                   # 4 waves per block
print("per-instance iters: mean %.0f max %.0f  max/mean %.3f"%(it.mean(),it.max(),it.max()/it.mean()))
print("per-wave (max of 4): mean %.0f max %.0f max/mean %.3f"%(w.mean(),w.max(),w.max()/w.mean()))
cu=b.max(axis=1).reshape(-1,2).max(axis=1)   # 2 blocks per CU (if blocks 2k,2k+1 share a CU: unknown) 
print("per-block max: mean %.0f max %.0f ratio %.3f"%(b.max(axis=1).mean(), b.max(axis=1).max(), b.max(axis=1).max()/b.max(axis=1).mean()))
q=np.percentile(w,[5,25,50,75,95,99]); print("wave percentiles", q)
blk=b.max(axis=1).reshape(32,16)   # [drive][tone]
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print("per drive (mean over tone):", blk.mean(axis=1))
print("per tone (mean over drive):", blk.mean(axis=0))
