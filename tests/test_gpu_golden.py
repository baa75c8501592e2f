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
