"""-m gpu: the one-launch CG iteration on the symmetric tile form (r06, csrc/pgo_sym_kernels.hip k_pipe_cg_sym): the owner-only
pipelined CG of the sharded path (Ghysels & Vanroose recurrences, one global reduction per iteration) with its product taken from the
form — every interior off-diagonal block read once — the eight vector recurrences of a tile's rows and the rows' Jacobi blocks in the
same launch.  It serves (a) every rank of a row-sharded solve above 600 k incidence slots (each rank keeps the form of ITS rows; cut
edges to other ranks are ghost columns), and (b) one rank above the universal stream's size limit (BASELINE configs[3] on one GPU).
PGO_SYM=1 forces the form on the small graphs the oracle can follow; PGO_NO_PIPELINE=1 keeps the universal stream out of the way on
one rank.  Held to the oracle's restatement of the same recurrences (pcg_form 1 there): same decisions, same CG count in every LM
iteration, costs to 1e-7; ranks bit-identical to each other; bit-reproducible."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(ds, info):
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    if info == "identity":
        g = ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, None)
    elif info == "full":
        rng = np.random.default_rng(12)
        A = rng.normal(size=(g.E, 6, 6))
        L = np.linalg.cholesky(A @ np.transpose(A, (0, 2, 1)) + 6.0 * np.eye(6)) * 0.6
        g = ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, L.reshape(-1, 36))
    return g


def _oracle(O, g, iters, cluster):
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    return O.solve(og, O.default_options(max_num_iterations=iters, linear_solver=1, pcg_cluster=cluster, pcg_form=1))


def _same_path(s, otr, rtol=1e-7):
    assert len(s.iterations) == len(otr)
    assert list(s.iterations["step_is_successful"]) == [int(x) for x in otr[:, 8]]
    assert list(s.iterations["linear_solver_iterations"]) == [int(x) for x in otr[:, 7]]
    assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=rtol)


@pytest.mark.parametrize("info", ["diag", "identity", "full"])
@pytest.mark.parametrize("cluster", [1, 2])
def test_one_rank_runs_the_pipelined_cg_on_the_form(gpu, ds, O, info, cluster, monkeypatch):
    monkeypatch.setenv("PGO_SYM", "1")
    monkeypatch.setenv("PGO_NO_PIPELINE", "1")
    g = _graph(ds, info)
    opt = dict(max_num_iterations=10, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster)
    runs = []
    for _ in range(2):
        prob, poses = gpu.problem_from_graph(g)
        runs.append((gpu.solve(gpu.SolverOptions(**opt), prob), poses))
    (s, poses), (s2, poses2) = runs
    assert s.cg_form == 2 and s.sym_form == 1                  # the one-launch iteration on the form really ran
    assert np.array_equal(poses, poses2) and np.array_equal(s.iterations["cost"], s2.iterations["cost"])
    op, osum, otr = _oracle(O, g, 10, cluster)
    _same_path(s, otr)
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-7)
    assert np.abs(poses - op).max() < 1e-5
    assert max(s.iterations["linear_solver_iterations"]) > 20
    # Ceres' refreshed CG stays available on the form (pcg_form 1: k_spmv_sym<0> + k_pcg_update)
    prob, _ = gpu.problem_from_graph(g)
    s1 = gpu.solve(gpu.SolverOptions(pcg_form=1, **opt), prob)
    assert s1.cg_form == 0 and s1.sym_form == 1
    assert list(s1.iterations["step_is_successful"]) == list(s.iterations["step_is_successful"])
    assert np.allclose(s1.iterations["cost"], s.iterations["cost"], rtol=1e-6)


def _virtual_ranks(gpu, g, world, opt_kw):
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = gpu.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            out[rank] = (gpu.solve(gpu.SolverOptions(**opt_kw), prob), poses)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    assert not errs, errs
    assert all(o is not None for o in out), "a virtual rank did not finish"
    gpu.loopback_destroy(group)
    return out


@pytest.mark.parametrize("world,direct", [(2, False), (3, False), (3, True), (8, False)])
@pytest.mark.parametrize("info", ["diag", "full"])
def test_sharded_ranks_keep_the_form_of_their_rows(gpu, ds, O, world, direct, info, monkeypatch):
    """Every rank of a row-sharded solve holds the symmetric form of ITS rows and runs k_pipe_cg_sym on it; the far ends of the edges
    that leave a rank are ghost columns filled from the exchange buffer.  direct: the exchange done by the kernels (peer table)."""
    monkeypatch.setenv("PGO_SYM", "1")
    monkeypatch.setenv("PGO_BLOCK", "256")          # (the incidence-slot partition that carries the first linearisation: whole pose pairs per work-group)
    if direct:
        monkeypatch.setenv("PGO_PEER_DIRECT", "1")
    g = _graph(ds, info)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    out = _virtual_ranks(gpu, g, world, opt)
    op, osum, otr = _oracle(O, g, 8, 2)
    for s, x in out:
        assert s.cg_form == 2 and s.sym_form == 1 and s.cg_exchange == (2 if direct else 3)      # (3: the host-enqueued exchange carries boundary rows only)
        _same_path(s, otr)
        assert np.array_equal(x, out[0][1])                     # all ranks bit-identical
        assert np.array_equal(s.iterations["cost"], out[0][0].iterations["cost"])
        assert np.abs(x - op).max() < 1e-5
    assert max(out[0][0].iterations["linear_solver_iterations"]) > 20
    # ... and the one-rank solve on the form (same kernel, one tile partition instead of `world`)
    monkeypatch.setenv("PGO_NO_PIPELINE", "1")
    prob, poses1 = gpu.problem_from_graph(g)
    one = gpu.solve(gpu.SolverOptions(**opt), prob)
    assert one.cg_form == 2 and one.sym_form == 1
    assert list(one.iterations["linear_solver_iterations"]) == list(out[0][0].iterations["linear_solver_iterations"])
    assert np.allclose(one.iterations["cost"], out[0][0].iterations["cost"], rtol=1e-8)
    assert np.abs(poses1 - out[0][1]).max() < 1e-6


@pytest.mark.parametrize("sym", [True, False])
def test_boundary_exchange_carries_the_same_bits_as_whole_segments(gpu, ds, monkeypatch, knobs, sym):
    """Sharded sessions (on the symmetric form, k_pipe_cg_sym, and on the incidence slots, k_pipe_cg), host-enqueued exchange: per CG
    iteration only the ranks' boundary rows (rows with an edge to another rank) + three sums each are all-gathered (Summary::cg_exchange
    3); with the knob shard_boundary = 0 the whole segments (1).  The kernels read the same numbers either way: every rank's poses and
    cost trace are bit-identical between the two."""
    monkeypatch.setenv("PGO_SYM", "1" if sym else "0")
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(3000, 15000, seed=9, loop_radius=6.0)      # (closures that reach across the shares: a few hundred boundary rows per rank)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    cuts, _ = gpu.row_shard_cuts(g.N, g.ia, g.ib, 4)
    owner = np.searchsorted(np.asarray(cuts[1:]), np.arange(g.N), side="right")
    assert (owner[g.ia] != owner[g.ib]).sum() > 50
    a = _virtual_ranks(gpu, g, 4, opt)
    prob, _ = gpu.problem_from_graph(g)              # (one rank: nothing to exchange)
    monkeypatch.delenv("PGO_SYM")
    prob.solver_begin(gpu.SolverOptions(**opt))
    assert prob.exchange_doubles() == 0
    prob.solver_end()
    monkeypatch.setenv("PGO_SYM", "1" if sym else "0")
    knobs(shard_boundary=0)
    b = _virtual_ranks(gpu, g, 4, opt)
    for (sa, xa), (sb, xb) in zip(a, b):
        assert sa.cg_exchange == 3 and sb.cg_exchange == 1 and sa.sym_form == (1 if sym else 0) and sb.sym_form == sa.sym_form
        assert np.array_equal(xa, xb) and np.array_equal(sa.iterations["cost"], sb.iterations["cost"])
        assert list(sa.iterations["linear_solver_iterations"]) == list(sb.iterations["linear_solver_iterations"])


def test_a_share_without_rows_keeps_the_ranks_off_the_symmetric_form(gpu, ds, monkeypatch):
    """Five poses on four ranks with the form forced (PGO_SYM=1): a rank without rows would launch an empty grid of tiles (r06,
    tools/fuzz_sharded.py seed 8078: "invalid configuration argument" on two ranks, the others waiting in a collective for ever) —
    every rank decides alike, from the shares, to stay on the incidence slots; and a rank whose solve fails releases an in-process group."""
    monkeypatch.setenv("PGO_SYM", "1")
    g = ds.manhattan_se3(5, 4, seed=3)          # (a chain: five poses, four odometry edges)
    opt = dict(max_num_iterations=6, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    out = _virtual_ranks(gpu, g, 4, opt)
    prob, x1 = gpu.problem_from_graph(g)
    one = gpu.solve(gpu.SolverOptions(**opt), prob)
    for s, x in out:
        assert s.sym_form == 0 and np.array_equal(x, out[0][1])
        assert np.allclose(s.iterations["cost"], one.iterations["cost"], rtol=1e-9, atol=1e-18)
