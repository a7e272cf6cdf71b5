"""Multi-GPU sharding of a sweep: one process per GPU, torch.distributed over RCCL/xGMI.

The hot path shards perfectly (instances never interact; time cannot be split), so there is
no collective on the data path.  What does cross xGMI:
  * ``broadcast_model``  -- rank 0's model block (matrices + element table, ~10 KB) to all
                            ranks before the run (``ncclBroadcast``),
  * ``gather_outputs``   -- optional collection of the sharded outputs on one rank
                            (grouped send/recv underneath ``dist.gather``),
  * ``reduce_reports``   -- failure/iteration counters (``ncclAllReduce``, a few words).
The reference has nothing comparable (single process, src/ACME.jl); this is the scale-out
the north star adds.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU
(used by the world_size-2 unit tests).
"""
from __future__ import annotations

import json

import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous instance range [lo, hi) of ``rank``; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_model(model, src=0, device=None):
    """Broadcast a DiscreteModel from ``src`` to every rank; returns it on all ranks."""
    import torch
    import torch.distributed as dist

    from .model import DiscreteModel
    rank = dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    if rank == src:
        payload = json.dumps(model.to_dict()).encode()
        n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    else:
        payload = b""
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, src)
    if rank == src:
        return model
    return DiscreteModel.from_dict(json.loads(bytes(buf.cpu().numpy().tobytes()).decode()))


def gather_outputs(y_local, counts, dst=0):
    """Collect the per-rank output shards [n_r, T, ny] on ``dst`` (None elsewhere).
    ``counts``: instances per rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    T, ny = y_local.shape[1], y_local.shape[2]
    if rank == dst:
        parts = [torch.empty((c, T, ny), dtype=y_local.dtype, device=y_local.device) for c in counts]
    else:
        parts = None
    if len(set(counts)) == 1:
        dist.gather(y_local.contiguous(), parts, dst=dst)
    else:  # ragged shards: point-to-point
        if rank == dst:
            reqs = []
            for r in range(world):
                if r == dst:
                    parts[r].copy_(y_local)
                else:
                    reqs.append(dist.irecv(parts[r], src=r))
            for q in reqs:
                q.wait()
        else:
            dist.send(y_local.contiguous(), dst=dst)
    return torch.cat(parts, dim=0) if rank == dst else None


def collect_outputs(y_local, mode="rank0", dst=0):
    """Collection of equally sized output shards [n, T, ny]: ``rank0`` -- every rank's shard to ``dst``
    (returns the [world*n, T, ny] tensor there, None elsewhere); ``allgather`` -- to every rank (one
    RCCL all-gather into a preallocated tensor).  With the gloo backend (CPU tests, one-device
    rehearsal) the shards travel as host tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and y_local.is_cuda:
        y_local = y_local.cpu()
    if mode == "allgather":
        out = torch.empty((world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype,
                          device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous())
        return out
    if mode == "rank0":
        return gather_outputs(y_local, [y_local.shape[0]] * world, dst=dst)
    raise ValueError(mode)


def reduce_reports(report_arrays, device=None):
    """All-reduce the solver counters of every rank: returns dict of global totals."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    s = torch.tensor([float(report_arrays["n_warn"].sum()),
                      float((report_arrays["first_nonfinite"] >= 0).sum()),
                      float(report_arrays["iters_total"].sum())], dtype=torch.float64, device=dev)
    m = torch.tensor([float(report_arrays["iters_max"].max(initial=0))], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return dict(n_warn=int(s[0]), n_nonfinite=int(s[1]), iters_total=int(s[2]), iters_max=int(m[0]))
