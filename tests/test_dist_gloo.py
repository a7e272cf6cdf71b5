"""World-size-2 tests of the sharding helpers on CPU (gloo): model broadcast, ragged output
gather, report reduction -- the N>1 path of bench.py without GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, load


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acme_jl_amd.dist import broadcast_model, gather_outputs, reduce_reports, shard_range
    from helpers import load as _load
    model = _load("superover_var") if rank == 0 else None
    model = broadcast_model(model, src=0)
    ref = _load("superover_var")
    same = all(np.array_equal(getattr(model, k), getattr(ref, k)) for k in ("a", "b", "c", "dy", "fy")) and \
        np.array_equal(model.subs[0].fq, ref.subs[0].fq) and model.subs[0].table == ref.subs[0].table and \
        model.subs[0].row_order == ref.subs[0].row_order
    n_total = 11                                    # ragged: 6 + 5
    lo, hi = shard_range(n_total, rank, world)
    y_local = torch.arange(lo, hi, dtype=torch.float64)[:, None, None].expand(hi - lo, 4, 1).contiguous()
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    y = gather_outputs(y_local, counts, dst=0)
    rep = dict(n_warn=np.array([rank + 1]), first_nonfinite=np.array([-1 if rank else 3]),
               iters_total=np.array([10 * (rank + 1)]), iters_max=np.array([7 + rank]))
    tot = reduce_reports(rep)
    # equally sized shards through both collection modes of bench.py --gather
    from acme_jl_amd.dist import collect_outputs
    shard = torch.full((3, 5, 2), float(rank + 1), dtype=torch.float64)
    g0 = collect_outputs(shard, mode="rank0", dst=0)
    ga = collect_outputs(shard, mode="allgather")
    expect = torch.cat([torch.full((3, 5, 2), float(r + 1), dtype=torch.float64) for r in range(world)])
    same = same and torch.equal(ga, expect) and ((g0 is None) if rank else torch.equal(g0, expect))
    if rank == 0:
        ok = same and y.shape == (n_total, 4, 1) and torch.equal(y[:, 0, 0], torch.arange(n_total, dtype=torch.float64))
        ok = ok and tot == dict(n_warn=3, n_nonfinite=1, iters_total=30, iters_max=8)
        out.put(bool(ok))
    else:
        out.put(bool(same))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_gather_reduce():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res)


def test_shard_range_covers_everything():
    from acme_jl_amd.dist import shard_range
    for n in (1, 7, 8192, 65536, 16385):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _hot_path_worker(rank, world, port, out):
    """One rank of the sharded HOT PATH: the kernel source itself (CPU wave-emulator library), instances
    shard_range(N, rank, world) of a superover sweep, model block broadcast from rank 0, outputs gathered on rank 0."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acme_jl_amd import runner
    from acme_jl_amd.dist import broadcast_model, gather_outputs, reduce_reports, shard_range
    from acme_jl_amd.model import CachingHomotopySolver
    from helpers import load as _load, sweep_inputs
    lib = runner.Library(os.path.join(ROOT, "tests", "emu", "libacme_emu.so"))
    N, T = 11, 160                                   # ragged shards: 6 + 5; two launches per rank
    u = sweep_inputs("superover_var", N, T, seed=3)  # every rank builds the same sweep and takes its slice
    model = broadcast_model(_load("superover_var", CachingHomotopySolver) if rank == 0 else None, src=0)
    lo, hi = shard_range(N, rank, world)
    r = runner.ModelRunner(model, hi - lo, lib=lib)
    y_local = np.concatenate([r.run(u[lo:hi, :, :100]), r.run(u[lo:hi, :, 100:])], axis=2)     # [n, ny, T]
    counts = [shard_range(N, k, world)[1] - shard_range(N, k, world)[0] for k in range(world)]
    y = gather_outputs(torch.from_numpy(np.ascontiguousarray(np.transpose(y_local, (0, 2, 1)))), counts, dst=0)
    tot = reduce_reports(r.report_arrays())
    if rank == 0:
        # the unsharded run of the same sweep, same library, same launch boundaries
        r1 = runner.ModelRunner(_load("superover_var", CachingHomotopySolver), N, lib=lib)
        y1 = np.concatenate([r1.run(u[:, :, :100]), r1.run(u[:, :, 100:])], axis=2)
        ra = r1.report_arrays()
        ok = np.array_equal(np.transpose(y.numpy(), (0, 2, 1)), y1)
        ok = ok and tot["iters_total"] == int(ra["iters_total"].sum()) and tot["n_warn"] == int(ra["n_warn"].sum())
        ok = ok and tot["iters_max"] == int(ra["iters_max"].max())
        out.put(bool(ok))
    else:
        out.put(True)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hot_path_matches_unsharded_run(emu_lib):
    """The hot path ITSELF sharded over two ranks (gloo): contiguous instance ranges, no data-path collective, model
    broadcast + ragged gather + counter reduction around it -- the gathered y is bit for bit the unsharded run's
    (an instance's arithmetic does not depend on which other instances share its launch), the reduced counters its sums."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_hot_path_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res)
