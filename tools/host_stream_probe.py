"""Streamed host-buffer runs of the headline workload: run time against the chunk size of the copy (ACME_HOST_STREAM_CHUNK) and
against the device-resident launch.   usage (GPU box): python tools/host_stream_probe.py [chunk ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from acme_jl_amd.model import CachingHomotopySolver  # noqa: E402
from acme_jl_amd.runner import ACME_MEM_HOST, ModelRunner  # noqa: E402
from helpers import load  # noqa: E402

N, T = 8192, 44100
m = load("superover_var", CachingHomotopySolver)
_, pots, amp = bench.grid_inputs("superover_grid", 0, 1, N, T)
u = bench.make_u(torch, torch.device("cuda"), m, pots, amp, N, T)
r = ModelRunner(m, N)
y = torch.empty((N, T, m.ny), dtype=torch.float64, device="cuda")
for _ in range(3):
    r.run_torch(u, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
r.run_torch(u, y)
torch.cuda.synchronize()
print("device-resident launch: %.1f ms (kernel %.1f)" % (1e3 * (time.perf_counter() - t0), r.last_kernel_ms()))
uh = u.cpu().numpy()
yh = np.empty((N, T, m.ny))
r.set_host_retention(True)
import ctypes as C  # noqa: E402
dp = C.POINTER(C.c_double)


def call():
    t0 = time.perf_counter()
    r.lib.check(r.lib.L.acme_batch_run(r.h, uh.ctypes.data_as(dp), yh.ctypes.data_as(dp), T, ACME_MEM_HOST, None))
    return 1e3 * (time.perf_counter() - t0)


call()
os.environ["ACME_HOST_STREAM_COPY_FIRST"] = "1"
print("copy first, then the streamed kernel: %.1f ms (kernel %.1f)" % (min(call() for _ in range(2)), r.last_kernel_ms()), flush=True)
os.environ.pop("ACME_HOST_STREAM_COPY_FIRST")
for chunk in (sys.argv[1:] or ["128", "64", "256", "32"]):
    if chunk.startswith("s"):          # s<n>: the fully staged pipeline with n time slices
        os.environ["ACME_HOST_ZEROCOPY"] = "0"
        os.environ["ACME_HOST_STAGED_SLICES"] = chunk[1:]
        print("staged, %4s slices: %.1f ms" % (chunk[1:], min(call() for _ in range(2))), flush=True)
        os.environ.pop("ACME_HOST_ZEROCOPY")
        continue
    os.environ["ACME_HOST_STREAM_CHUNK"] = chunk
    print("chunk %4s: %.1f ms (kernel %.1f)" % (chunk, min(call() for _ in range(2)), r.last_kernel_ms()), flush=True)
