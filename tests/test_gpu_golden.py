"""-m gpu: the HIP path against the committed fixtures (tests/golden/, oracle outputs + reference data)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_edge_vectors(gpu, ds):
    v = np.load(os.path.join(G, "edge_vectors.npz"))
    n = len(v["pa"])
    poses = np.zeros((2 * n, 7))
    poses[0::2, :3], poses[0::2, 3:] = v["pa"], v["qa"]
    poses[1::2, :3], poses[1::2, 3:] = v["pb"], v["qb"]
    g = ds.PoseGraphData(poses, np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2), np.concatenate([v["mp"], v["mq"]], axis=1),
                         v["L"].reshape(n, 36))
    prob, _ = gpu.problem_from_graph(g, loss=gpu.TRIVIAL, constant_first=False)
    cost, r, ja, jb, grad = prob.evaluate()
    for ref in ("analytic", "autodiff"):
        assert np.abs(r - v["r_" + ref]).max() <= 1e-11 * np.abs(v["r_" + ref]).max()
        assert np.abs(ja - v["ja_" + ref]).max() <= 1e-11 * np.abs(v["ja_" + ref]).max()
        assert np.abs(jb - v["jb_" + ref]).max() <= 1e-11 * np.abs(v["jb_" + ref]).max()
    assert cost == pytest.approx(0.5 * (v["r_analytic"] ** 2).sum(), rel=1e-12)


def test_toy_graph_trace(gpu, ds):
    t = np.load(os.path.join(G, "toy_graph.npz"))
    g = ds.PoseGraphData(t["poses"], t["ia"], t["ib"], t["meas"], t["sqrt_info"])
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=100, function_tolerance=1e-12,
                                    linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    tr = t["trace"]
    assert s.initial_cost == pytest.approx(float(t["initial_cost"]), rel=1e-12)
    n = min(len(tr), len(s.iterations))
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in tr[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], tr[:n, 1], rtol=1e-8)
    assert s.final_cost == pytest.approx(float(t["final_cost"]), rel=1e-9)
    assert np.abs(poses - t["final_poses"]).max() <= 1e-6


def test_kitti00_replay_pcg_same_path(gpu, ds, O):
    """C1 graph (4541 poses / 5179 edges, identity information = the kernels' fast path) with truncated PCG:
    same policy on both sides, so the LM paths coincide."""
    k = np.load(os.path.join(G, "kitti00.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=15, linear_solver_type=gpu.BLOCK_JACOBI_PCG), prob)
    og = O.Graph(k["origin"], k["ia"], k["ib"], k["meas"], None)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=15, linear_solver=1))
    n = min(len(otr), len(s.iterations), 8)
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-6)
    assert np.array_equal(poses[0], k["origin"][0])


def test_kitti00_replay_reference_options_exact(gpu, ds, O):
    """The reference's own configuration (finial.cpp:534-536: SPARSE_NORMAL_CHOLESKY, 1000 iterations, defaults)
    on the C1 graph: exact steps on the GPU (block-sparse Cholesky, nested dissection) against the oracle's
    exact steps.  Same iteration count, same cost trace, same stopping reason; and at tight convergence the
    poses agree (pose parity is only meaningful there)."""
    k = np.load(os.path.join(G, "kitti00.npz"))
    tr = np.load(os.path.join(G, "kitti00_trace.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    assert s.linear_solver_used == 0 and s.factor_nnz_blocks > 4541
    ref = tr["trace_default"]
    assert len(s.iterations) == len(ref)
    assert list(s.iterations["step_is_successful"]) == [int(x) for x in ref[:, 8]]
    assert np.allclose(s.iterations["cost"], ref[:, 1], rtol=1e-7)
    assert s.final_cost == pytest.approx(float(tr["final_cost_default"]), rel=1e-8)
    assert "Function tolerance" in s.message
    assert np.abs(poses[:, :3] - tr["poses_default"][:, :3]).max() < 1e-4
    # tight convergence
    prob2, poses2 = gpu.problem_from_graph(g)
    s2 = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY,
                                     function_tolerance=1e-15, parameter_tolerance=1e-13), prob2)
    assert s2.final_cost == pytest.approx(float(tr["final_cost_tight"]), rel=1e-10)
    assert np.abs(poses2[:, :3] - tr["poses_tight"][:, :3]).max() < 1e-5
    # information only: distance to the reference's committed output (cannot match: stand-in loop measurements)
    d = np.linalg.norm(poses[:, :3] - k["updated"][:, :3], axis=1)
    assert np.isfinite(d).all()


def test_kitti00_dense_candidates_config3(gpu, ds, O):
    """BASELINE.json configs[2]: every id of Edge_Candidates_index.txt becomes an edge (4541 poses /
    20 499 edges), synthetic measurements (seed 20260929): final trajectory vs the CPU oracle at tight convergence."""
    k = np.load(os.path.join(G, "kitti00.npz"))
    cands = {int(kk): k["cand_flat"][k["cand_offsets"][i]:k["cand_offsets"][i + 1]].tolist() for i, kk in enumerate(k["cand_keys"])}
    g = ds.graph_from_candidates(k["origin"], cands, seed=20260929)
    assert g.N == 4541 and g.E == 20499
    prob, poses = gpu.problem_from_graph(g)
    opt = dict(max_num_iterations=100, function_tolerance=1e-13, parameter_tolerance=1e-11)
    s = gpu.solve(gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, **opt), prob)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, None)
    op, osum, otr = O.solve(og, O.default_options(linear_solver=0, **opt))
    assert s.is_solution_usable()
    assert s.linear_solver_used == 0 and s.factor_nnz_blocks > 60000     # a GPU factorisation (multifrontal by default), not the PCG stand-in
    assert s.initial_cost == pytest.approx(osum.initial_cost, rel=1e-12)
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-9)
    assert np.abs(poses[:, :3] - op[:, :3]).max() < 1e-5            # metres
    dq = np.minimum(np.abs(poses[:, 3:] - op[:, 3:]).max(axis=1), np.abs(poses[:, 3:] + op[:, 3:]).max(axis=1))
    assert dq.max() < 1e-6
    assert np.array_equal(poses[0], g.poses[0])


def test_reference_g2o_excerpt_solve(gpu, ds, O):
    """Lines of the reference's own 00.g2o (tests/golden/g2o_00_excerpt.g2o): read, perturb the vertices, solve with the
    reference's options on the GPU and with the oracle — same trace, same poses."""
    g = ds.read_g2o(os.path.join(G, "g2o_00_excerpt.g2o"))
    rng = np.random.default_rng(7)
    start = g.poses.copy()
    start[1:, :3] += rng.normal(0, 0.2, (g.N - 1, 3))
    start[1:, 3:] = ds.qmul(ds.qexp_half(rng.normal(0, 0.02, (g.N - 1, 3))), start[1:, 3:])
    h = ds.PoseGraphData(start, g.ia, g.ib, g.meas, None)
    prob, poses = gpu.problem_from_graph(h)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    op, osum, otr = O.solve(O.Graph(start, g.ia, g.ib, g.meas, None), O.default_options(max_num_iterations=1000))
    assert s.num_iterations == osum.num_iterations and s.termination_type == gpu.CONVERGENCE
    assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=1e-7, atol=1e-12)
    assert np.abs(poses - op).max() < 1e-6
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-7)
    assert s.final_cost < 1e-3 * s.initial_cost      # (an open chain: the far end is weakly constrained, poses are not compared to the file)
