"""CPU, world_size 2 over gloo: the multi-process decomposition of the path.
Each rank owns a contiguous share of the pose rows (pgo_row_shard_cuts: the rule pgo_comm_init itself applies — shares cut where the incidence slots balance) and every edge incident to them;
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
    cut, seg = pkg.row_shard_cuts(g.N, g.ia, g.ib, world)        # THE ownership rule (r06: shares cut where the incidence slots balance)
    lo, hi = cut[rank], cut[rank + 1]
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
            b0, b1 = cut[k], cut[k + 1]
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


def _pipe_worker(rank, world, port, out):
    """The sharded path's owner-only pipelined CG (pgo_kernels.hip k_pipe_cg) restated in numpy over gloo:
    every rank multiplies ITS rows of A = J'J + D, updates the eight vectors of its rows only, applies its own 6 x 6 Jacobi blocks, and
    ONE all-gather per iteration carries [m of the owned rows | (r,u), (w,u), x'(b + r)] — the product's exchange layout
    (pgo_row_shard_cuts segments).  Every rank derives alpha, beta and the Q-tolerance stop from the same gathered numbers; the
    iterates must be those of standard preconditioned CG on the whole system."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgo_loader
    pkg = pgo_loader.load()
    ds = pgo_loader.datasets()
    from oracle import oracle as O
    g = ds.manhattan_se3(240, 800, seed=33)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, r, ja, jb = O.evaluate(og)
    n = 6 * g.N
    # whole system (every rank builds it only to take ITS rows and, at the end, the single-process truth)
    A = np.zeros((n, n)); b = np.zeros(n)
    for e in range(g.E):
        a, c = 6 * int(g.ia[e]), 6 * int(g.ib[e])
        J = np.zeros((6, n)); J[:, a:a + 6] = ja[e]; J[:, c:c + 6] = jb[e]
        A += J.T @ J; b -= J.T @ r[e]
    d = np.diag(A).copy()
    d[d == 0.0] = 1e4                      # (the constant first pose: its columns are zero, the block becomes the identity)
    A += 1e-4 * np.diag(d)
    cut, seg = pkg.row_shard_cuts(g.N, g.ia, g.ib, world)
    lo, hi = cut[rank], cut[rank + 1]
    rows = slice(6 * lo, 6 * hi)
    Aown = A[rows]                                                     # the owned block rows
    Minv = [np.linalg.inv(A[6 * v:6 * v + 6, 6 * v:6 * v + 6]) for v in range(lo, hi)]
    prec = lambda v: np.concatenate([Minv[i] @ v[6 * i:6 * i + 6] for i in range(hi - lo)])

    def exchange(m_own, sums):                                         # [world][seg * 6 + 4], in place in the product
        buf = torch.zeros(seg * 6 + 4, dtype=torch.float64)
        buf[: m_own.size] = torch.from_numpy(m_own)
        buf[seg * 6: seg * 6 + 3] = torch.tensor(sums, dtype=torch.float64)
        outs = [torch.zeros(seg * 6 + 4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, buf)
        full, tot = [], np.zeros(3)
        for k in range(world):
            b0, b1 = cut[k], cut[k + 1]
            full.append(outs[k][: 6 * (b1 - b0)].numpy())
            tot += outs[k][seg * 6: seg * 6 + 3].numpy()               # rank order: the same bits on every rank
        return np.concatenate(full), tot

    bo = b[rows]
    x = np.zeros(6 * (hi - lo)); rr = bo.copy(); u = prec(rr)
    ufull, _ = exchange(u, [0, 0, 0])
    w = Aown @ ufull
    m = prec(w)
    mfull, (gamma, delta, qsum) = exchange(m, [rr @ u, w @ u, 0.0])
    z = np.zeros_like(x); q = np.zeros_like(x); s_ = np.zeros_like(x); p = np.zeros_like(x)
    gamma_prev = alpha_prev = q_prev = 0.0
    its = 0
    for it in range(200):
        Q1 = -qsum
        if it > 0 and it * (Q1 - q_prev) / Q1 < 0.1:
            break
        beta = gamma / gamma_prev if it > 0 else 0.0
        alpha = gamma / (delta - beta * gamma / alpha_prev) if it > 0 else gamma / delta
        nn = Aown @ mfull
        z = nn + beta * z; q = m + beta * q; s_ = w + beta * s_; p = u + beta * p
        x = x + alpha * p; rr = rr - alpha * s_; u = u - alpha * q; w = w - alpha * z
        m = prec(w)
        gamma_prev, alpha_prev, q_prev = gamma, alpha, Q1
        mfull, (gamma, delta, qsum) = exchange(m, [rr @ u, w @ u, x @ (bo + rr)])
        its += 1
    xfull, _ = exchange(x, [0, 0, 0])
    # single-process standard PCG with the same stop rule
    Mi = [np.linalg.inv(A[6 * v:6 * v + 6, 6 * v:6 * v + 6]) for v in range(g.N)]
    P_ = lambda v: np.concatenate([Mi[i] @ v[6 * i:6 * i + 6] for i in range(g.N)])
    xs = np.zeros(n); rs = b.copy(); zs = P_(rs); ps = zs.copy(); rho = rs @ zs; Q0 = 0.0; k = 0
    for k in range(1, 201):
        qs = A @ ps; al = rho / (ps @ qs)
        xs = xs + al * ps; rs = rs - al * qs
        Qk = -xs @ (b + rs)
        if k > 1 and k * (Qk - Q0) / Qk < 0.1:
            break
        Q0 = Qk
        zs = P_(rs); rho_new = rs @ zs; ps = zs + (rho_new / rho) * ps; rho = rho_new
    ok = its == k and np.allclose(xfull, xs, rtol=1e-9, atol=1e-12 * np.abs(xs).max())
    if rank == 0:
        out.put((bool(ok), its, k, float(np.abs(xfull - xs).max() / np.abs(xs).max())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_owner_only_pipelined_cg_world2():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    ok, its, k, err = out.get()
    assert ok, (its, k, err)
    assert its > 5
