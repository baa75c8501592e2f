"""-m gpu: the row-sharded path.  A single-GPU box cannot host two RCCL ranks, so the sharding logic is
validated with the loopback transport (virtual ranks = host threads of one process, device-to-device segment copies),
and the RCCL transport is exercised at world size 1 (same call sequence, ncclAllGather inside the captured CG batch)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve_sharded(pkg, g, world, opt_kw):
    group = pkg.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = pkg.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            s = pkg.solve(pkg.SolverOptions(**opt_kw), prob)
            out[rank] = (s, poses)
        except Exception as e:  # a failing rank would leave the others waiting: surface it
            errs.append(e)

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errs, errs
    assert all(o is not None for o in out), "a virtual rank did not finish"
    pkg.loopback_destroy(group)
    return out


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("cluster", [1, 2])
def test_loopback_ranks_match_single_rank(gpu, ds, world, cluster):
    g = ds.manhattan_se3(1001, 3700, seed=31)
    opt = dict(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster)
    prob, poses = gpu.problem_from_graph(g)
    ref = gpu.solve(gpu.SolverOptions(**opt), prob)
    out = _solve_sharded(gpu, g, world, opt)
    for s, p in out:
        # same decisions, same CG iteration counts; sums are folded in a different order -> rounding-level differences
        assert list(s.iterations["step_is_successful"]) == list(ref.iterations["step_is_successful"])
        assert list(s.iterations["linear_solver_iterations"]) == list(ref.iterations["linear_solver_iterations"])
        assert np.allclose(s.iterations["cost"], ref.iterations["cost"], rtol=1e-9)
        assert np.abs(p - poses).max() < 1e-7
    # every rank holds the identical result
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("cluster", [1, 2])
def test_owner_only_cg_and_replicated_cg_agree(gpu, ds, cluster, monkeypatch):
    """Several ranks run the truncated CG in its owner-only pipelined form (k_pipe_cg: one launch and one all-gather per iteration,
    every rank updates its own rows) unless PGO_SHARD_PIPE=0 keeps the standard form with the vector update
    replicated on every rank.  Same CG iteration counts in every LM iteration — runs of 100+ iterations included —, same decisions,
    costs to 1e-8, and both equal to one rank."""
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster)
    prob, poses = gpu.problem_from_graph(g)
    ref = gpu.solve(gpu.SolverOptions(**opt), prob)
    res = {}
    monkeypatch.setenv("PGO_BLOCK", "256")      # (work-groups of 256 slots: every pose pair 2i, 2i+1 fits one — what the owner-only form needs)
    for form in ("1", "0"):
        monkeypatch.setenv("PGO_SHARD_PIPE", form)
        res[form] = _solve_sharded(gpu, g, 4, opt)
    for form, out in res.items():
        for s, p in out:
            assert s.cg_form == (2 if form == "1" else 1)      # the form that really ran (Summary::cg_form)
            assert list(s.iterations["step_is_successful"]) == list(ref.iterations["step_is_successful"]), form
            assert list(s.iterations["linear_solver_iterations"]) == list(ref.iterations["linear_solver_iterations"]), form
            assert np.allclose(s.iterations["cost"], ref.iterations["cost"], rtol=1e-8), form
            assert np.abs(p - poses).max() < 1e-6, form
        assert all(np.array_equal(out[0][1], o[1]) for o in out)
    assert max(ref.iterations["linear_solver_iterations"]) > 20


def test_linear_solve_on_loopback_ranks_returns_the_whole_solution(gpu, ds):
    """pgo_linear_solve hands every rank the whole x: it keeps the standard CG (vector update on every rank) whatever form the
    Levenberg-Marquardt loop of the same problem takes, and equals the single-rank solve."""
    g = ds.manhattan_se3(901, 3300, seed=12)
    opt = gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=1e-3, max_linear_solver_iterations=300)
    rng = np.random.default_rng(3)
    d2 = np.full(6 * g.N, 1e2)       # (well conditioned: the truncated CG stops by its Q-tolerance long before the iteration limit)
    b = rng.normal(size=6 * g.N)
    prob, _ = gpu.problem_from_graph(g)
    x1, it1 = prob.linear_solve(d2, b, opt)
    world = 3
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            p, _ = gpu.problem_from_graph(g)
            p.comm_init_loopback(group, rank)
            out[rank] = p.linear_solve(d2, b, opt)
        except Exception as e:  # a failing rank would leave the others waiting: surface it
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    gpu.loopback_destroy(group)
    for x, it in out:
        assert it == it1 < 300 and np.allclose(x, x1, rtol=1e-7, atol=1e-9 * np.abs(x1).max())
        assert np.array_equal(x, out[0][0])


def test_linear_solve_behind_an_owner_only_solve_on_the_same_problems(gpu, ds, monkeypatch):
    """r04 advisor: an LM session in the owner-only CG leaves `the last linearisation exchanged only diagonals` behind; pgo_linear_solve
    linearises by itself (full exchange) and runs the replicated standard CG — it must not trip over the stale flag."""
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(901, 3300, seed=12)
    lm_opt = gpu.SolverOptions(max_num_iterations=4, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    ls_opt = gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=1e-3, max_linear_solver_iterations=300)
    rng = np.random.default_rng(3)
    d2, b = np.full(6 * g.N, 1e2), rng.normal(size=6 * g.N)
    world = 2
    group = gpu.loopback_create(world)
    out, forms, errs = [None] * world, [None] * world, []

    def run(rank):
        try:
            p, _ = gpu.problem_from_graph(g)
            p.comm_init_loopback(group, rank)
            forms[rank] = gpu.solve(lm_opt, p).cg_form
            out[rank] = p.linear_solve(d2, b, ls_opt)
        except Exception as e:
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    gpu.loopback_destroy(group)
    assert forms == [2, 2]                                   # the LM session really ran the owner-only form
    assert out[0][1] == out[1][1] and np.array_equal(out[0][0], out[1][0])


def test_loopback_exact_request_falls_back_to_tight_pcg(gpu, ds, O):
    g = ds.manhattan_se3(300, 1000, seed=4)
    opt = dict(max_num_iterations=15, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    out = _solve_sharded(gpu, g, 2, opt)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=15, linear_solver=0))
    s, p = out[0]
    assert s.linear_solver_used == 2            # factorisation needs every row: sharded runs use PCG to 1e-13
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-6)


def test_rccl_transport_at_world_one(gpu, ds):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather (forced at world 1), inside the captured CG batch."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(600, 2200, seed=5)
opt = dict(max_num_iterations=10, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
prob, poses = pkg.problem_from_graph(g)
ref = pkg.solve(pkg.SolverOptions(**opt), prob)
prob2, poses2 = pkg.problem_from_graph(g)
prob2.comm_init(pkg.comm_unique_id(), 0, 1)
s = pkg.solve(pkg.SolverOptions(**opt), prob2)
assert list(s.iterations["linear_solver_iterations"]) == list(ref.iterations["linear_solver_iterations"])
assert np.array_equal(poses, poses2), np.abs(poses - poses2).max()
print("RCCL_OK", s.final_cost)
''' % ROOT
    env = dict(os.environ, PGO_FORCE_EXCHANGE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "RCCL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_device_initiated_exchange_equals_the_all_gather(gpu, ds, world, monkeypatch):
    """The owner-only CG with the exchange done by the kernels themselves (every producing launch stores its segment into every
    rank's buffer; its one-work-group tail signals a per-rank flag and waits for everybody's: DeviceGraph::peer_tab) against the
    host-enqueued all-gather between the launches: same kernels, same arithmetic, only the transport differs — bit-identical
    iteration records and poses.  (Loopback ranks: peers' device pointers are valid in this process; the RCCL transport keeps
    the all-gather.)"""
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    monkeypatch.setenv("PGO_BLOCK", "256")
    monkeypatch.setenv("PGO_PEER_DIRECT", "0")
    ref = _solve_sharded(gpu, g, world, opt)
    monkeypatch.setenv("PGO_PEER_DIRECT", "1")
    out = _solve_sharded(gpu, g, world, opt)
    for (s, p), (s0, p0) in zip(out, ref):
        assert s.termination_type != gpu.FAILURE and s.cg_form == 2
        for f in ("step_is_successful", "linear_solver_iterations", "cost", "trust_region_radius"):
            assert np.array_equal(s.iterations[f], s0.iterations[f]), f
        assert np.array_equal(p, p0)
