import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from acme_jl_amd.model import DiscreteModel, CachingHomotopySolver
from acme_jl_amd.runner import ModelRunner
N,T=8192,4410
fixture,pots,amp=bench.grid_inputs("superover_grid",0,1,N,T)
m=DiscreteModel.load(os.path.join("tests","golden",fixture+".json"), solver=CachingHomotopySolver)
u=np.zeros((N,4,T)); u[:,0]=np.sin(2*np.pi*1000/44100*np.arange(T)); u[:,1:]=pots[:,:,None]
r=ModelRunner(m,N)
y=r.run(u)
t0=time.perf_counter()
for _ in range(3): y=r.run(u)
dt=(time.perf_counter()-t0)/3
print("host-buffer run! (numpy in/out, incl. layout transposes + H2D + kernel + D2H): %.1f ms per %d samples -> %.3g inst*samples/s; kernel alone %.1f ms"%(dt*1e3,T,N*T/dt,r.last_kernel_ms()))
# the C ABI's own cost (what a Julia caller pays: its arrays already are [N][T][nu]): no numpy transposes
import ctypes as C
from acme_jl_amd.runner import ACME_MEM_HOST
ub = np.ascontiguousarray(np.transpose(u, (0, 2, 1)))
yb = np.empty((N, T, m.ny))
def raw():
    r.lib.check(r.lib.L.acme_batch_run(r.h, ub.ctypes.data_as(C.POINTER(C.c_double)), yb.ctypes.data_as(C.POINTER(C.c_double)), T, ACME_MEM_HOST, None))
raw()
t0 = time.perf_counter()
for _ in range(3): raw()
dt = (time.perf_counter() - t0) / 3
print("acme_batch_run(ACME_MEM_HOST) alone: %.1f ms per %d samples -> %.3g inst*samples/s (H2D %.2f GB, D2H %.2f GB)" % (dt * 1e3, T, N * T / dt, ub.nbytes / 1e9, yb.nbytes / 1e9))
