"""-m gpu tests of pose / landmark problems (the half r02 left out; role of g2o's BlockSolver,
Thirdparty/g2o/g2o/core/block_solver.hpp:47-87: Hpp / Hll / Hpl and the Schur complement onto the poses).  A 3-D point is a size-3
Euclidean block in caller memory (pgo_problem_add_point), an observation the point in the observing pose's frame
(pgo_problem_add_point_observation_batch); the exact solvers eliminate the point blocks first — the Schur complement — and order
the poses by nested dissection of the reduced graph.  The checker is the CPU oracle on the same problem stated as one pose graph
(points = nodes with a constant identity quaternion, observations = between-factors without rotation information)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(gpu, pl):
    g = pl.graph
    poses = np.array(g.poses, dtype=np.float64, order="C", copy=True)
    points = np.array(pl.points, dtype=np.float64, order="C", copy=True)
    p = gpu.Problem()
    assert p.add_poses(poses) == 0
    p.add_se3_between(g.ia, g.ib, g.meas, g.sqrt_info)
    first = p.add_points(points)
    assert first == g.N and p.num_poses == g.N + len(points)
    p.add_point_observations(pl.obs_pose, first + pl.obs_point, pl.obs_z, pl.obs_sqrt_info3)
    p.set_loss(gpu.HUBER, 1.0)
    p.set_pose_constant(0, 3)
    return p, poses, points


@pytest.mark.parametrize("exact", [True, False])
def test_pose_landmark_solve_matches_oracle(gpu, O, ds, exact):
    """200 poses / 2000 points / ~15 000 observations: the LM trace of the GPU solve (exact steps: points eliminated first;
    PCG: 12x12 clusters) equals the oracle's — same accept / reject sequence, same stopping reason, costs 1e-8 (PCG 1e-6) — and the
    caller's pose and point arrays hold the oracle's result to 1e-6 (PCG: 3e-4)."""
    pl = ds.pose_landmark_toy(n_poses=200, n_points=2000, seed=20260932)
    full, cmask = pl.as_pose_graph()
    og = O.Graph(full.poses, full.ia, full.ib, full.meas, full.sqrt_info, cmask)
    ls, cl = (gpu.SPARSE_NORMAL_CHOLESKY, 1) if exact else (gpu.BLOCK_JACOBI_PCG, 2)
    p, poses, points = _build(gpu, pl)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=60, linear_solver_type=ls, pcg_cluster_poses=cl), p)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=60, linear_solver=0 if exact else 1, pcg_cluster=cl))
    assert len(s.iterations) == len(otr) and len(otr) > 6
    assert list(s.iterations["step_is_successful"]) == [int(v) for v in otr[:, 8]]
    assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=1e-8 if exact else 1e-6)     # (truncated CG: the steps agree to the CG's own rounding, 8e-9 measured)
    assert s.termination_type == osum.termination_type and s.reason == osum.reason
    N = pl.graph.N
    tol = 1e-6 if exact else 3e-4      # (truncated PCG: 1.1e-4 measured in the flat directions at costs equal to 1e-6)
    assert np.abs(poses - op[:N]).max() <= tol and np.abs(points - op[N:, :3]).max() <= tol
    assert np.abs(points - pl.truth_points).max() < np.abs(pl.points - pl.truth_points).max() * 0.2      # the points did move towards the truth
    if exact:
        assert s.linear_solver_used == 0 and s.num_linear_solver_iterations == 0
    else:
        assert s.num_linear_solver_iterations == osum.num_linear_iterations


def test_points_first_is_an_ordering_not_another_answer(gpu, ds):
    """The same problem handed over as a plain pose graph (no point blocks declared: plain nested dissection) gives the same LM
    trace to rounding; declared points are eliminated first and the factor is no larger."""
    pl = ds.pose_landmark_toy(n_poses=200, n_points=2000, seed=20260932)
    full, cmask = pl.as_pose_graph()
    opt = dict(max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    p, poses, points = _build(gpu, pl)
    a = gpu.solve(gpu.SolverOptions(**opt), p)
    q, qposes = gpu.problem_from_graph(full)
    for v in np.nonzero(cmask)[0]:
        q.set_pose_constant(int(v), int(cmask[v]))
    b = gpu.solve(gpu.SolverOptions(**opt), q)
    assert len(a.iterations) == len(b.iterations)
    assert list(a.iterations["step_is_successful"]) == list(b.iterations["step_is_successful"])
    assert np.allclose(a.iterations["cost"], b.iterations["cost"], rtol=1e-9)
    N = pl.graph.N
    assert np.abs(poses - qposes[:N]).max() <= 1e-7 and np.abs(points - qposes[N:, :3]).max() <= 1e-7
    assert np.array_equal(qposes[N:, 3:], np.tile([0.0, 0.0, 0.0, 1.0], (len(points), 1)))
    print("factor blocks: points first %d, plain nested dissection %d" % (a.factor_nnz_blocks, b.factor_nnz_blocks))
    assert a.factor_nnz_blocks <= b.factor_nnz_blocks


def test_point_api_errors(gpu, ds):
    pl = ds.pose_landmark_toy(n_poses=20, n_points=30, seed=3)
    p, poses, points = _build(gpu, pl)
    with pytest.raises(gpu.PgoError):
        p.add_point_observations([25], [3], [[0.0, 0.0, 0.0]])          # 25 is a point, not a pose
    with pytest.raises(gpu.PgoError):
        p.add_point_observations([3], [4], [[0.0, 0.0, 0.0]])           # 4 is a pose, not a point
    assert p.add_points(points) == 20                                      # same memory again: the same blocks
    assert p.num_poses == 50


def test_pose_landmark_pcg_sharded_over_loopback_ranks(gpu, ds):
    """The same pose / landmark problem with its rows (poses first, then the points) sharded over three loopback ranks: the
    owner-only CG of the sharded path takes the single-rank decisions with the single-rank CG counts, and every
    rank ends with the same poses and points."""
    import threading
    pl = ds.pose_landmark_toy(n_poses=120, n_points=900, seed=20260933)
    opt = dict(max_num_iterations=20, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    p0, poses0, points0 = _build(gpu, pl)
    ref = gpu.solve(gpu.SolverOptions(**opt), p0)
    world = 3
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            p, poses, points = _build(gpu, pl)
            p.comm_init_loopback(group, rank)
            out[rank] = (gpu.solve(gpu.SolverOptions(**opt), p), poses, points)
        except Exception as e:   # a failing rank would leave the others waiting: surface it
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    assert all(o is not None for o in out)
    gpu.loopback_destroy(group)
    for s, poses, points in out:
        assert list(s.iterations["step_is_successful"]) == list(ref.iterations["step_is_successful"])
        assert list(s.iterations["linear_solver_iterations"]) == list(ref.iterations["linear_solver_iterations"])
        assert np.allclose(s.iterations["cost"], ref.iterations["cost"], rtol=1e-8)
        assert np.abs(poses - poses0).max() < 1e-6 and np.abs(points - points0).max() < 1e-6
        assert np.array_equal(poses, out[0][1]) and np.array_equal(points, out[0][2])
