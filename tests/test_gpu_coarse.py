"""-m gpu: the PCG's COARSE LEVEL (r06, csrc/pgo_coarse.hip; options.pcg_coarse_aggregate): M^-1 = M_J^-1 + P (P'AP)^-1 P' with the 2-pose
cluster Jacobi and an aggregation coarse space of rigid-body modes.  Measured in the oracle first (tests/test_oracle_pcg_forms.py,
tools/two_level_oracle.py); here the HIP path is held to the oracle's restatement of the same preconditioner in the same pipelined
recurrences: same accept / reject decisions, CG counts within one iteration (the oracle solves the coarse system by Cholesky, the
kernels apply an explicit inverse: the last bits of a coarse correction differ), costs to 1e-5 — and to what the coarse level is FOR:
the truncated PCG (eta = 0.1) reaches the exact path's cost."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("agg,info", [(32, "diag"), (64, "diag"), (32, "identity"), (48, "full")])
def test_coarse_level_matches_the_oracles_two_level_pcg(gpu, ds, O, agg, info):
    g = ds.manhattan_se3(1200, 4800, seed=5)
    if info == "identity":
        g = ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, None)
    elif info == "full":
        rng = np.random.default_rng(12)
        A = rng.normal(size=(g.E, 6, 6))
        g = ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, (np.linalg.cholesky(A @ np.transpose(A, (0, 2, 1)) + 6.0 * np.eye(6)) * 0.6).reshape(-1, 36))
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    its = 30
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg,
                                    eta=0.1, max_linear_solver_iterations=500), prob)
    assert s.coarse_level == (g.N + agg - 1) // agg and s.cg_form == 2
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=its, linear_solver=1, pcg_cluster=-agg, pcg_form=1, eta=0.1, max_linear_solver_iterations=500))
    n = min(len(s.iterations), len(otr))
    assert n >= 20
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
    dcg = np.abs(np.asarray(s.iterations["linear_solver_iterations"][:n], dtype=int) - otr[:n, 7].astype(int))
    assert dcg.max() <= 1 and (dcg == 0).mean() >= 0.8, dcg
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-5)          # (an LM iteration whose CG stopped one iteration apart: 1e-6 near convergence)
    # bit-reproducible
    prob2, poses2 = gpu.problem_from_graph(g)
    s2 = gpu.solve(gpu.SolverOptions(max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg,
                                     eta=0.1, max_linear_solver_iterations=500), prob2)
    assert np.array_equal(poses, poses2) and np.array_equal(s.iterations["cost"], s2.iterations["cost"])


def test_coarse_level_reaches_the_exact_paths_cost(gpu, ds):
    """What it is for: eta = 0.1 from dead reckoning to its own stop ends at the exact path's cost in a few hundred CG iterations; the
    cluster Jacobi alone is still above it after many times the CG work."""
    g = ds.manhattan_se3(1200, 4800, seed=5)
    prob, _ = gpu.problem_from_graph(g)
    exact = gpu.solve(gpu.SolverOptions(max_num_iterations=400, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    prob, _ = gpu.problem_from_graph(g)
    two = gpu.solve(gpu.SolverOptions(max_num_iterations=400, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=32, eta=0.1), prob)
    prob, _ = gpu.problem_from_graph(g)
    jac = gpu.solve(gpu.SolverOptions(max_num_iterations=two.num_iterations - 1, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=0.1), prob)
    assert two.final_cost == pytest.approx(exact.final_cost, rel=1e-4)
    assert two.num_linear_solver_iterations <= 1000 and jac.num_linear_solver_iterations >= 3 * two.num_linear_solver_iterations
    assert jac.final_cost > exact.final_cost * (1.0 + 1e-4)


def test_coarse_level_is_refused_where_it_cannot_run(gpu, ds):
    g = ds.manhattan_se3(300, 900, seed=1)
    prob, _ = gpu.problem_from_graph(g)
    with pytest.raises(gpu.PgoError):
        gpu.solve(gpu.SolverOptions(max_num_iterations=3, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, pcg_coarse_aggregate=32), prob)
    with pytest.raises(gpu.PgoError):
        gpu.solve(gpu.SolverOptions(max_num_iterations=3, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_coarse_aggregate=4), prob)


def _virtual_ranks(gpu, g, world, opt_kw):
    import threading
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
        t.join(240)
    assert not errs, errs
    assert all(o is not None for o in out), "a virtual rank did not finish"
    gpu.loopback_destroy(group)
    return out


@pytest.mark.parametrize("world,direct", [(2, False), (3, True), (8, False)])
def test_coarse_level_on_row_sharded_ranks(gpu, ds, world, direct, monkeypatch):
    """Several ranks: aggregates never straddle ranks; every rank forms the Galerkin row panels of ITS aggregates and restricts
    over ITS rows, the panels (per LM iteration) and the restricted vector (per CG iteration, 6 doubles per aggregate) are all-gathered,
    the inverse is replicated.  All ranks bit-identical; the truncated PCG ends at the exact path's cost with the CG work of the one-rank
    solve (other aggregate boundaries: not the same counts).  direct: a peer table, where one was set up, is set aside for the session."""
    if direct:
        monkeypatch.setenv("PGO_PEER_DIRECT", "1")
    g = ds.manhattan_se3(1200, 4800, seed=5)
    agg = 32
    opt = dict(max_num_iterations=400, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg, eta=0.1)
    out = _virtual_ranks(gpu, g, world, opt)
    prob, _ = gpu.problem_from_graph(g)
    one = gpu.solve(gpu.SolverOptions(**opt), prob)
    prob, _ = gpu.problem_from_graph(g)
    exact = gpu.solve(gpu.SolverOptions(max_num_iterations=400, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    s0, x0 = out[0]
    for s, x in out:
        assert s.cg_form == 2 and s.cg_exchange == 1 and s.coarse_level >= (g.N + agg - 1) // agg
        assert np.array_equal(x, x0) and np.array_equal(s.iterations["cost"], s0.iterations["cost"])
    assert s0.final_cost == pytest.approx(exact.final_cost, rel=1e-4)
    assert s0.num_linear_solver_iterations <= 1.5 * one.num_linear_solver_iterations + 50


def test_coarse_level_on_ranks_at_a_size_where_launches_overlap(gpu, ds):
    """20 000 poses on 4 virtual ranks: a coarse matrix of 2 184 unknowns inverted by every rank while the others' kernels share the device.
    (r06: the block Gauss-Jordan update read the old pivot column panel from the matrix another work-group of the same strip rewrites — it
    went unnoticed while a launch had the device to itself; the panel is copied by the pivot kernel now.)"""
    g = ds.manhattan_se3(20000, 100000, seed=5)
    opt = dict(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=64, eta=0.1)
    out = _virtual_ranks(gpu, g, 4, opt)
    prob, _ = gpu.problem_from_graph(g)
    one = gpu.solve(gpu.SolverOptions(**opt), prob)
    s0, x0 = out[0]
    for s, x in out:
        assert np.array_equal(x, x0) and np.array_equal(s.iterations["cost"], s0.iterations["cost"])
    assert s0.final_cost == pytest.approx(one.final_cost, rel=2e-2)
    assert abs(s0.num_linear_solver_iterations - one.num_linear_solver_iterations) <= 0.25 * one.num_linear_solver_iterations


@pytest.mark.parametrize("world", [2, 3, 5])
def test_coarse_level_on_ranks_matches_the_oracle_with_the_same_segments(gpu, ds, O, world):
    """The sharded solve against the oracle's two-level PCG with aggregates formed inside the same row shares (pgo_row_shard_cuts ->
    oracle.set_coarse_cuts): the same preconditioner in the same recurrences — same accept / reject decisions, CG counts within one
    iteration, costs to 1e-5, as on one rank."""
    g = ds.manhattan_se3(1200, 4800, seed=5)
    agg, its = 32, 30
    cuts, rows_per = gpu.row_shard_cuts(g.N, g.ia, g.ib, world)
    out = _virtual_ranks(gpu, g, world, dict(max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg,
                                             eta=0.1, max_linear_solver_iterations=500))
    s = out[0][0]
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    O.set_coarse_cuts(cuts)
    try:
        op, osum, otr = O.solve(og, O.default_options(max_num_iterations=its, linear_solver=1, pcg_cluster=-agg, pcg_form=1, eta=0.1, max_linear_solver_iterations=500))
    finally:
        O.set_coarse_cuts(None)
    n = min(len(s.iterations), len(otr))
    assert n >= 20
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
    dcg = np.abs(np.asarray(s.iterations["linear_solver_iterations"][:n], dtype=int) - otr[:n, 7].astype(int))
    assert dcg.max() <= 1 and (dcg == 0).mean() >= 0.8, dcg
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-5)
    assert np.abs(out[0][1] - op).max() < 1e-3
