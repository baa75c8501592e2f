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
