"""CPU, world_size 2 over gloo: the multi-process decomposition of the path.
Each rank owns a contiguous share of the pose rows (pgo_row_shard_range: the rule pgo_comm_init itself applies) and every edge incident to them;
the per-rank pieces of J'r, of the diagonal J'J blocks and of one block SpMV, all-gathered over gloo,
must equal the single-process result.  The arithmetic here is the CPU oracle's (no GPU in this
container); the partition logic is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgo_loader
    pkg = pgo_loader.load()
    ds = pgo_loader.datasets()
    from oracle import oracle as O
    g = ds.manhattan_se3(300, 1000, seed=21)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, r, ja, jb = O.evaluate(og)
    lo, hi, seg = pkg.row_shard_range(g.N, rank, world)
    # rows owned by this rank: gradient and diagonal blocks from every incident edge (cut edges are
    # evaluated by both owners, never exchanged)
    grad = np.zeros((hi - lo, 6))
    diag = np.zeros((hi - lo, 6, 6))
    x = np.random.default_rng(0).normal(size=(g.N, 6))       # replicated vector
    y = np.zeros((hi - lo, 6))
    for e in range(g.E):
        a, b = int(g.ia[e]), int(g.ib[e])
        if lo <= a < hi:
            grad[a - lo] += ja[e].T @ r[e]
            diag[a - lo] += ja[e].T @ ja[e]
            y[a - lo] += ja[e].T @ (ja[e] @ x[a] + jb[e] @ x[b])
        if lo <= b < hi:
            grad[b - lo] += jb[e].T @ r[e]
            diag[b - lo] += jb[e].T @ jb[e]
            y[b - lo] += jb[e].T @ (ja[e] @ x[a] + jb[e] @ x[b])
    # one all-gather per operator application (padded equal-size segments, as RCCL all-gather needs)
    def gather(local, width):
        buf = torch.zeros(seg * width, dtype=torch.float64)
        buf[: local.size] = torch.from_numpy(local.reshape(-1))
        outs = [torch.zeros(seg * width, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, buf)
        parts = []
        for k in range(world):
            b0, b1, _ = pkg.row_shard_range(g.N, k, world)
            parts.append(outs[k][: (b1 - b0) * width].numpy().reshape(b1 - b0, width))
        return np.concatenate(parts)
    full_grad, full_y = gather(grad, 6), gather(y, 6)
    full_diag = gather(diag, 36).reshape(g.N, 6, 6)
    # single-process truth
    tg, td, ty = np.zeros((g.N, 6)), np.zeros((g.N, 6, 6)), np.zeros((g.N, 6))
    for e in range(g.E):
        a, b = int(g.ia[e]), int(g.ib[e])
        jx = ja[e] @ x[a] + jb[e] @ x[b]
        tg[a] += ja[e].T @ r[e]; tg[b] += jb[e].T @ r[e]
        td[a] += ja[e].T @ ja[e]; td[b] += jb[e].T @ jb[e]
        ty[a] += ja[e].T @ jx; ty[b] += jb[e].T @ jx
    ok = (np.array_equal(full_grad, tg) and np.array_equal(full_diag, td) and np.array_equal(full_y, ty))
    # cost: edge-sharded partial sums + all-reduce
    e0, e1 = pkg.shard_range(g.E, rank, world)
    sub = O.Graph(g.poses, g.ia[e0:e1], g.ib[e0:e1], g.meas[e0:e1], g.sqrt_info[e0:e1])
    c = torch.tensor([O.cost(sub)], dtype=torch.float64)
    dist.all_reduce(c)
    ok = ok and abs(float(c) - O.cost(og)) <= 1e-12 * O.cost(og)
    if rank == 0:
        out.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sharded_reduction_world2():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert out.get() is True
