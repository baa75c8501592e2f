"""-m gpu parity tests: the HIP path (through the C ABI of include/pgo.h) against the CPU oracle on the
same seeded inputs.  Tolerances are stated per test; everything is FP64."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_graph(ds, n, e, seed, info="full", unit=True):
    """Random poses / measurements (large residuals, random 6x6 sqrt information)."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((n, 7))
    poses[:, :3] = rng.normal(0, 2.0, size=(n, 3))
    q = rng.normal(size=(n, 4))
    poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    if not unit:
        poses[:, 3:] *= (1.0 + 1e-4 * rng.normal(size=(n, 1)))
    ia = rng.integers(0, n, size=e).astype(np.int32)
    ib = (ia + 1 + rng.integers(0, n - 1, size=e)).astype(np.int32) % n
    meas = np.zeros((e, 7))
    meas[:, :3] = rng.normal(0, 1.0, size=(e, 3))
    mq = rng.normal(size=(e, 4))
    meas[:, 3:] = mq / np.linalg.norm(mq, axis=1, keepdims=True)
    sqrt_info = None
    if info == "full":
        A = rng.normal(size=(e, 6, 6))
        M = A @ np.transpose(A, (0, 2, 1)) + 6 * np.eye(6)
        sqrt_info = np.linalg.cholesky(M).reshape(e, 36) * 0.3
    elif info == "diag":
        sqrt_info = np.repeat(np.diag(rng.uniform(0.5, 3.0, size=6)).reshape(1, 36), e, axis=0)
    return ds.PoseGraphData(poses, ia, ib, meas, sqrt_info)


def _pair(gpu, O, g, cmask=None, loss=1, loss_a=1.0):
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info, cmask)
    prob, poses = gpu.problem_from_graph(g, loss=loss, loss_a=loss_a, constant_first=False)
    cm = og.cmask
    for v in np.nonzero(cm)[0]:
        prob.set_pose_constant(int(v), int(cm[v]))
    return prob, poses, og


@pytest.mark.parametrize("info", ["full", "diag", None])
@pytest.mark.parametrize("loss", [0, 1])
def test_evaluate_matches_oracle(gpu, O, ds, info, loss):
    g = _random_graph(ds, 200, 900, seed=11, info=info, unit=(info != "diag"))
    cmask = np.zeros(200, dtype=np.uint8)
    cmask[0] = 3
    cmask[5] = 1
    cmask[9] = 2
    prob, poses, og = _pair(gpu, O, g, cmask, loss=loss, loss_a=1.5)
    cost, r, ja, jb, grad = prob.evaluate()
    ocost, orr, oja, ojb = O.evaluate(og, loss_kind=loss, loss_a=1.5)
    assert cost == pytest.approx(ocost, rel=1e-12)
    scale = max(1.0, np.abs(oja).max())
    assert np.abs(r - orr).max() <= 1e-11 * max(1.0, np.abs(orr).max())
    assert np.abs(ja - oja).max() <= 1e-11 * scale
    assert np.abs(jb - ojb).max() <= 1e-11 * scale
    # gradient = sum J^T r
    og_grad = np.zeros((g.N, 6))
    np.add.at(og_grad, g.ia, np.einsum("eki,ek->ei", oja, orr))
    np.add.at(og_grad, g.ib, np.einsum("eki,ek->ei", ojb, orr))
    assert np.abs(grad - og_grad).max() <= 1e-10 * max(1.0, np.abs(og_grad).max())


@pytest.mark.parametrize("info", ["full", None])
def test_normal_equations_match_oracle(gpu, O, ds, info):
    g = _random_graph(ds, 60, 400, seed=5, info=info)
    cmask = np.zeros(60, dtype=np.uint8)
    cmask[0] = 3
    cmask[7] = 2
    prob, poses, og = _pair(gpu, O, g, cmask)
    diag, off, grad = prob.normal_equations()
    ocost, H, ograd = O.normal_equations_dense(og)
    tol = 1e-10 * np.abs(H).max()
    for v in range(g.N):
        assert np.abs(diag[v] - H[6 * v:6 * v + 6, 6 * v:6 * v + 6]).max() <= tol
    # off-diagonal: oracle accumulates duplicates of a pair; compare sums per (a,b)
    acc = {}
    for e in range(g.E):
        key = (int(g.ia[e]), int(g.ib[e]))
        acc[key] = acc.get(key, 0) + off[e]
    merged = {}
    for (a, b), blk in acc.items():
        k2 = (a, b) if a < b else (b, a)
        merged[k2] = merged.get(k2, 0) + (blk if a < b else blk.T)
    for (a, b), blk in merged.items():
        assert np.abs(blk - H[6 * a:6 * a + 6, 6 * b:6 * b + 6]).max() <= tol
    assert np.abs(grad.reshape(-1) - ograd).max() <= 1e-10 * max(1.0, np.abs(ograd).max())


def test_plus_matches_oracle(gpu, O, ds):
    g = _random_graph(ds, 300, 400, seed=3)
    prob, poses, og = _pair(gpu, O, g, np.zeros(300, dtype=np.uint8))
    rng = np.random.default_rng(0)
    delta = rng.normal(0, 0.3, size=(300, 6))
    delta[10, 3:] = 0.0  # zero rotation step keeps q bitwise
    before = poses.copy()
    prob.plus(delta)
    for v in range(300):
        assert np.allclose(poses[v, :3], before[v, :3] + delta[v, :3], rtol=0, atol=1e-15)
        assert np.allclose(poses[v, 3:], O.quat_plus(before[v, 3:], delta[v, 3:]), rtol=0, atol=1e-15)
    assert np.array_equal(poses[10, 3:], before[10, 3:])


def test_linear_solve_matches_exact_oracle(gpu, O, ds):
    g = ds.manhattan_se3(150, 500, seed=2)
    prob, poses, og = _pair(gpu, O, g)
    rng = np.random.default_rng(1)
    d2 = rng.uniform(0.1, 1.0, size=g.N * 6)
    b = rng.normal(size=g.N * 6)
    b[:6] = 0.0  # constant pose
    x, it = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
    xo, _ = O.linear_solve(og, d2, b, linear_solver=0)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    # truncated PCG with the Ceres Q-tolerance: same iteration count and iterate as the oracle's PCG
    x2, it2 = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=0.1))
    xo2, ito2 = O.linear_solve(og, d2, b, linear_solver=1, q_tol=0.1, max_it=500)
    assert it2 == ito2
    assert np.abs(x2 - xo2).max() <= 1e-9 * np.abs(xo2).max()


@pytest.mark.parametrize("solver", ["pcg", "exact"])
def test_lm_trace_matches_oracle(gpu, O, ds, solver):
    g = ds.manhattan_se3(300, 1000, seed=4)
    prob, poses, og = _pair(gpu, O, g)
    ls = gpu.BLOCK_JACOBI_PCG if solver == "pcg" else gpu.SPARSE_NORMAL_CHOLESKY
    opt = gpu.SolverOptions(max_num_iterations=40, linear_solver_type=ls)
    s = gpu.solve(opt, prob)
    oposes, osum, otrace = O.solve(og, O.default_options(max_num_iterations=40, linear_solver=1 if solver == "pcg" else 0))
    assert s.initial_cost == pytest.approx(osum.initial_cost, rel=1e-12)
    n = min(len(s.iterations), len(otrace), 12)
    # same accept/reject decisions and costs over the first iterations (later ones amplify rounding)
    assert list(s.iterations["step_is_successful"][:n]) == [int(v) for v in otrace[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], otrace[:n, 1], rtol=1e-7)
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-5)
    assert s.is_solution_usable()
    # constant pose untouched, result written in place
    assert np.array_equal(poses[0], g.poses[0])
    assert not np.array_equal(poses[1:], g.poses[1:])


def test_tight_convergence_pose_parity(gpu, O, ds):
    """Both sides run to tight convergence (function_tolerance -> 0): poses must agree."""
    g = ds.manhattan_se3(200, 800, seed=9, sigma_t=0.02, sigma_r=0.004)
    prob, poses, og = _pair(gpu, O, g, loss=0)
    opt = gpu.SolverOptions(max_num_iterations=200, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY,
                            function_tolerance=1e-14, parameter_tolerance=1e-12)
    s = gpu.solve(opt, prob)
    oposes, osum, _ = O.solve(og, O.default_options(max_num_iterations=200, linear_solver=0, loss_kind=0,
                                                   function_tolerance=1e-14, parameter_tolerance=1e-12))
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-9)
    assert np.abs(poses[:, :3] - oposes[:, :3]).max() <= 1e-6          # metres
    dq = np.minimum(np.abs(poses[:, 3:] - oposes[:, 3:]).max(axis=1), np.abs(poses[:, 3:] + oposes[:, 3:]).max(axis=1))
    assert dq.max() <= 1e-7


def test_edge_cases(gpu, O, ds):
    # empty edge set: cost 0, converges immediately, poses untouched
    poses = np.zeros((3, 7))
    poses[:, 6] = 1.0
    p = gpu.Problem()
    p.add_poses(poses)
    p.set_pose_constant(0)
    s = gpu.solve(gpu.SolverOptions(), p)
    assert s.final_cost == 0.0 and s.is_solution_usable()
    # a star: one hub with > block incidences (fat row spanning several chunks)
    rng = np.random.default_rng(2)
    n = 700
    truth = np.zeros((n, 7))
    truth[:, :3] = rng.normal(0, 3, size=(n, 3))
    q = rng.normal(size=(n, 4))
    truth[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    ia = np.arange(1, n, dtype=np.int32)
    ib = np.zeros(n - 1, dtype=np.int32)
    meas = ds.relative_pose(truth[ia], truth[ib])
    meas[:, :3] += rng.normal(0, 0.01, size=(n - 1, 3))
    init = truth.copy()
    init[1:, :3] += rng.normal(0, 0.1, size=(n - 1, 3))
    g = ds.PoseGraphData(init, ia, ib, meas, None)
    prob, poses2, og = _pair(gpu, O, g)
    diag, off, grad = prob.normal_equations()
    ocost, H, ograd = O.normal_equations_dense(og)
    assert np.abs(diag[0] - H[:6, :6]).max() <= 1e-9 * np.abs(H[:6, :6]).max()
    assert np.abs(grad.reshape(-1) - ograd).max() <= 1e-9 * max(1.0, np.abs(ograd).max())
    s2 = gpu.solve(gpu.SolverOptions(max_num_iterations=20, linear_solver_type=gpu.BLOCK_JACOBI_PCG), prob)
    _, osum, _ = O.solve(og, O.default_options(max_num_iterations=20, linear_solver=1))
    assert s2.final_cost == pytest.approx(osum.final_cost, rel=1e-6)


def test_api_misuse_reports_errors(gpu):
    p = gpu.Problem()
    poses = np.zeros((2, 7))
    poses[:, 6] = 1
    p.add_poses(poses)
    with pytest.raises(gpu.PgoError):
        p.add_se3_between([0], [5], np.zeros((1, 7)))       # unknown pose
    with pytest.raises(gpu.PgoError):
        p.add_se3_between([1], [1], np.zeros((1, 7)))       # self edge
    with pytest.raises(gpu.PgoError):
        p.set_loss(gpu.HUBER, -1.0)


@pytest.mark.parametrize("cluster", [2, 4])
def test_cluster_jacobi_pcg_matches_oracle(gpu, O, ds, cluster):
    """Cluster-Jacobi preconditioner (12x12 / 24x24 chain blocks): same iterates and iteration counts as the
    oracle's PCG with the same blocks; the exact solution is unchanged."""
    g = ds.manhattan_se3(301, 1100, seed=12)          # 301: the last cluster is partial
    prob, poses, og = _pair(gpu, O, g)
    rng = np.random.default_rng(1)
    d2 = rng.uniform(0.01, 0.5, size=g.N * 6)
    b = rng.normal(size=g.N * 6)
    b[:6] = 0.0
    x, it = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=0.1, pcg_cluster_poses=cluster))
    xo, ito = O.linear_solve(og, d2, b, linear_solver=100 + cluster, q_tol=0.1, max_it=500)
    assert it == ito
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    x1, it1 = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=0.1, pcg_cluster_poses=1))
    assert it <= it1
    # LM with the cluster preconditioner follows the oracle's LM with the same preconditioner
    opt = gpu.SolverOptions(max_num_iterations=25, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster)
    s = gpu.solve(opt, prob)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=25, linear_solver=1, pcg_cluster=cluster))
    n = min(len(s.iterations), len(otr), 10)
    assert list(s.iterations["linear_solver_iterations"][:n]) == [int(v) for v in otr[:n, 7]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-5)


@pytest.mark.parametrize("kind,a", [(2, 1.3), (3, 0.8), (4, 2.0), (5, 5.0)])
def test_other_ceres_losses(gpu, O, ds, kind, a):
    """SoftLOne / Cauchy / Arctan / switchable constraint in closed form: evaluation and LM trace against the oracle."""
    g = _random_graph(ds, 150, 600, seed=21, info="diag")
    prob, poses, og = _pair(gpu, O, g, loss=kind, loss_a=a)
    cost, r, ja, jb, grad = prob.evaluate()
    ocost, orr, oja, ojb = O.evaluate(og, loss_kind=kind, loss_a=a)
    assert cost == pytest.approx(ocost, rel=1e-12)
    assert np.abs(r - orr).max() <= 1e-11 * max(1.0, np.abs(orr).max())
    assert np.abs(ja - oja).max() <= 1e-11 * max(1.0, np.abs(oja).max())
    g2 = ds.manhattan_se3(200, 700, seed=6)
    prob2, poses2, og2 = _pair(gpu, O, g2, loss=kind, loss_a=a)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=15, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob2)
    op, osum, otr = O.solve(og2, O.default_options(max_num_iterations=15, linear_solver=0, loss_kind=kind, loss_a=a))
    n = min(len(s.iterations), len(otr), 8)
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)


@pytest.mark.parametrize("cluster", [1, 2, 4])
def test_awkward_topologies_match_oracle(gpu, O, ds, cluster):
    """Duplicate edges, an isolated free pose, partial constancy (p only / q only), non-unit quaternions, pose count
    not a multiple of the preconditioner cluster: same LM trace as the oracle with the same policy."""
    g = _random_graph(ds, 37, 90, seed=11, info="diag", unit=False)
    ia = np.concatenate([g.ia, g.ia[:7]])                 # duplicates of the first seven edges
    ib = np.concatenate([g.ib, g.ib[:7]])
    keep = (ia != 36) & (ib != 36)                        # pose 36 keeps no edge at all
    g = ds.PoseGraphData(g.poses, ia[keep], ib[keep], np.concatenate([g.meas, g.meas[:7]])[keep],
                         np.concatenate([g.sqrt_info, g.sqrt_info[:7]])[keep])
    cmask = np.zeros(37, dtype=np.uint8)
    cmask[0], cmask[5], cmask[9] = 3, 1, 2
    prob, poses, og = _pair(gpu, O, g, cmask)
    before = poses.copy()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster), prob)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=12, linear_solver=1, pcg_cluster=cluster))
    n = min(len(otr), len(s.iterations))
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)
    assert np.array_equal(poses[0], before[0]) and np.array_equal(poses[36], before[36])   # constant / unconstrained
    assert np.array_equal(poses[5, :3], before[5, :3]) and np.array_equal(poses[9, 3:], before[9, 3:])
    assert not np.array_equal(poses[5, 3:], before[5, 3:]) and not np.array_equal(poses[9, :3], before[9, :3])
    assert np.abs(poses - op).max() < 1e-6


def test_degenerate_problems_terminate_like_ceres(gpu, ds):
    g = _random_graph(ds, 6, 10, seed=3, info=None)
    # every parameter block constant: nothing to optimise, solution usable, poses untouched
    prob, poses = gpu.problem_from_graph(g, constant_first=False)
    for v in range(6):
        prob.set_pose_constant(v)
    before = poses.copy()
    s = gpu.solve(gpu.SolverOptions(), prob)
    assert s.is_solution_usable() and s.num_iterations <= 1 and np.array_equal(poses, before)
    assert s.final_cost == pytest.approx(s.initial_cost, rel=1e-15)
    # a non-finite pose: evaluation fails, the solve reports FAILURE and leaves the parameters alone
    bad = g.poses.copy()
    bad[2, 1] = np.nan
    prob2, poses2 = gpu.problem_from_graph(ds.PoseGraphData(bad, g.ia, g.ib, g.meas, None))
    before2 = poses2.copy()
    s2 = gpu.solve(gpu.SolverOptions(), prob2)
    assert s2.termination_type == gpu.FAILURE and not s2.is_solution_usable()
    assert np.array_equal(poses2, before2, equal_nan=True)


def test_incremental_problem_growth(gpu, O, ds):
    """The reference builds its graph once, but the C ABI allows adding poses / edges between solves (SLAM back-ends
    do): a problem grown in two steps and solved twice ends where a problem built in one go ends."""
    g = ds.manhattan_se3(400, 1200, seed=5)
    n1 = 250
    e1 = np.nonzero((g.ia < n1) & (g.ib < n1))[0]
    e2 = np.nonzero(~((g.ia < n1) & (g.ib < n1)))[0]
    poses = g.poses.copy()
    p = gpu.Problem()
    p.add_poses(poses[:n1])
    p.set_loss(gpu.HUBER, 1.0)
    p.set_pose_constant(0)
    p.add_se3_between(g.ia[e1], g.ib[e1], g.meas[e1], g.sqrt_info[e1])
    opt = gpu.SolverOptions(max_num_iterations=40, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, function_tolerance=1e-12)
    s1 = gpu.solve(opt, p)
    assert s1.num_poses == n1 and s1.final_cost < s1.initial_cost
    first = poses[:n1].copy()
    p.add_poses(poses[n1:])
    p.add_se3_between(g.ia[e2], g.ib[e2], g.meas[e2], g.sqrt_info[e2])
    s2 = gpu.solve(opt, p)
    assert s2.num_poses == g.N and s2.num_edges == g.E
    # the same state reached by a problem built in one go from the intermediate poses
    start = g.poses.copy()
    start[:n1] = first
    order = np.concatenate([e1, e2])
    h = ds.PoseGraphData(start, g.ia[order], g.ib[order], g.meas[order], g.sqrt_info[order])
    q, qposes = gpu.problem_from_graph(h)
    s3 = gpu.solve(opt, q)
    assert s2.initial_cost == pytest.approx(s3.initial_cost, rel=1e-13)
    assert s2.final_cost == pytest.approx(s3.final_cost, rel=1e-10)
    assert np.abs(poses - qposes).max() < 1e-8


def test_direct_solver_with_dense_separators(gpu, O, ds, monkeypatch):
    """A lattice walk with many loop closures: the elimination tree ends in dense separator chains, which the enumerated
    6x6 factorisation (pinned here with PGO_FRONT=0; by default this graph goes to the multifrontal solver,
    tests/test_gpu_front.py) processes as SPLIT levels and PANEL steps.  Same solution as the
    oracle's exact solve, and the LM run reports the factorisation (not the PCG stand-in) as the solver used."""
    monkeypatch.setenv("PGO_FRONT", "0")
    g = ds.manhattan_se3(2000, 8000, seed=3)
    prob, poses, og = _pair(gpu, O, g)
    rng = np.random.default_rng(5)
    d2 = rng.uniform(0.05, 2.0, size=g.N * 6)
    b = rng.normal(size=g.N * 6)
    b[:6] = 0.0
    x, it = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
    xo, _ = O.linear_solve(og, d2, b, linear_solver=0)
    assert it == 0
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=5, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    _, osum, otr = O.solve(og, O.default_options(max_num_iterations=5, linear_solver=0))
    assert s.linear_solver_used == 0 and s.factor_levels > 100
    n = min(len(otr), len(s.iterations))
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-8)


def test_exact_request_served_by_pcg_converges(gpu, O, ds, monkeypatch):
    """An exact request whose factorisation is declined (here: switched off) is served by PCG run to a 1e-13 relative
    residual of the recurrence.  It must reach the exact solve's answer and STOP: with Ceres' periodic r = b - Ax refresh
    applied in this mode the 1e-13 test never fires on an ill-conditioned graph (regression: 27x the iterations)."""
    monkeypatch.setenv("PGO_NO_DIRECT", "1")
    g = ds.manhattan_se3(2000, 8000, seed=3)
    prob, poses, og = _pair(gpu, O, g)
    rng = np.random.default_rng(5)
    d2 = rng.uniform(0.05, 2.0, size=g.N * 6)
    b = rng.normal(size=g.N * 6)
    b[:6] = 0.0
    x, it = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
    xo, _ = O.linear_solve(og, d2, b, linear_solver=0)
    assert 0 < it < 2000
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=5, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    _, osum, otr = O.solve(og, O.default_options(max_num_iterations=5, linear_solver=0))
    assert s.linear_solver_used == 2
    assert s.num_linear_solver_iterations < 5000 * max(1, s.num_iterations)
    n = min(len(otr), len(s.iterations))
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)


@pytest.mark.parametrize("budget", [None, "100000", "40"])
def test_exact_request_with_per_iteration_choice(gpu, O, ds, monkeypatch, knobs, budget):
    """Factorisation admitted but above the always-direct budget (forced here): every LM iteration is served either by
    the factorisation or by PCG to 1e-13, an over-budget PCG try is redone with the factorisation.  Whatever the mix
    (default budget: mostly factorisations; huge: PCG after the first; tiny: every try over budget), the LM trace is
    the oracle's exact-solve trace."""
    monkeypatch.setenv("PGO_FRONT", "0")                      # the per-iteration choice belongs to the enumerated factorisation
    monkeypatch.setenv("PGO_DIRECT_MAX_STEPS", "10")
    if budget:
        knobs(hybrid_budget=int(budget))
    g = ds.manhattan_se3(2000, 8000, seed=3)
    prob, poses, og = _pair(gpu, O, g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=8, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    _, osum, otr = O.solve(og, O.default_options(max_num_iterations=8, linear_solver=0))
    assert s.linear_solver_used == 3 and s.factor_nnz_blocks > 2000
    solves = len(s.iterations) - 1                           # iteration 0 has no linear solve
    if budget == "100000":
        assert s.num_linear_solver_iterations > 0 and 1 <= s.num_factorizations < solves   # PCG did serve iterations
    if budget == "40":
        assert s.num_factorizations == solves                # every PCG try over budget, every iteration factorised
    n = min(len(otr), len(s.iterations))
    assert n == len(otr) == len(s.iterations)
    assert list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)


def test_evaluate_special_configurations(gpu, O, ds):
    """Hand-picked edge states: zero residual, identical poses, antipodal quaternion representatives, half-turn relative
    rotation, non-unit quaternions, large translations; plus Plus() with a zero rotation step (the sin(x)/x branch)."""
    s2 = np.sqrt(0.5)
    poses = np.array([
        [0, 0, 0, 0, 0, 0, 1.0],
        [1, 2, 3, 0, 0, s2, s2],
        [1, 2, 3, 0, 0, -s2, -s2],          # same rotation as pose 1, other representative
        [-4, 0.5, 2, 1, 0, 0, 0.0],         # half turn about x
        [1e4, -2e4, 3e3, 0.1, 0.2, 0.3, 0.9],   # far away, not unit
        [0, 0, 0, 0, 0, 0, 1.0],            # identical to pose 0
    ])
    ia = np.array([0, 1, 2, 3, 4, 5, 1, 3], dtype=np.int32)
    ib = np.array([1, 2, 3, 4, 5, 0, 0, 0], dtype=np.int32)
    unit = poses.copy()
    unit[:, 3:] /= np.linalg.norm(unit[:, 3:], axis=1, keepdims=True)
    meas = ds.relative_pose(unit[ia], unit[ib])
    meas[1] = [0, 0, 0, 0, 0, 0, 1.0]        # identity measurement between two representatives of one pose
    meas[5] = [0, 0, 0, 0, 0, 0, -1.0]       # identical poses, measurement = -identity quaternion
    meas[6, :3] += 0.25                      # a genuinely non-zero residual
    meas[7, 3:] = [s2, 0, s2, 0.0]
    g = ds.PoseGraphData(poses, ia, ib, meas, None)
    for loss in (0, 1):
        prob, p2, og = _pair(gpu, O, g, loss=loss)
        cost, r, ja, jb, grad = prob.evaluate()
        ocost, orr, oja, ojb = O.evaluate(og, loss_kind=loss)
        scale = max(1.0, np.abs(oja).max())
        assert np.abs(r - orr).max() <= 1e-9 * max(1.0, np.abs(orr).max())      # poses 1e4 m away: absolute cancellation error
        assert np.abs(ja - oja).max() <= 1e-11 * scale and np.abs(jb - ojb).max() <= 1e-11 * scale
        assert cost == pytest.approx(ocost, rel=1e-10, abs=1e-12)
    # Plus with zero rotation increments on some poses and zero translation on others
    prob, p2, og = _pair(gpu, O, g)
    d = np.zeros((6, 6))
    d[1, :3] = [0.1, -0.2, 0.3]
    d[2, 3:] = [0.0, 0.0, 1e-12]
    d[3, 3:] = [0.4, 0.0, -0.1]
    expect = p2.copy()
    for v in range(6):
        expect[v, :3] += d[v, :3]
        expect[v, 3:] = O.quat_plus(expect[v, 3:], d[v, 3:])
    prob.plus(d)
    assert np.abs(p2 - expect).max() <= 1e-15 * 3e4
    assert np.array_equal(p2[0], poses[0]) and np.array_equal(p2[1, 3:], poses[1, 3:])


def test_evaluation_entry_points_are_refused_during_a_session(gpu, ds):
    """pgo_evaluate / pgo_normal_equations / pgo_linear_solve / pgo_plus would overwrite the device-resident LM state (pose
    ping-pong, Jacobi scaling, linearisation): between solver_begin and solver_end they return an error, and the session
    continues unharmed (same result as an uninterrupted one)."""
    g = ds.manhattan_se3(300, 1000, seed=4)
    opt = gpu.SolverOptions(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG)
    prob, poses = gpu.problem_from_graph(g)
    ref = gpu.solve(opt, prob)
    prob2, poses2 = gpu.problem_from_graph(g)
    prob2.solver_begin(opt)
    prob2.solver_step(3)
    for call in (prob2.evaluate, prob2.normal_equations, lambda: prob2.plus(np.zeros((g.N, 6))),
                 lambda: prob2.linear_solve(np.ones(6 * g.N), np.zeros(6 * g.N))):
        with pytest.raises(gpu.PgoError) as ei:
            call()
        assert "solver session" in str(ei.value)
    prob2.solver_step(100)
    s2 = prob2.solver_end()
    assert s2.final_cost == ref.final_cost and np.array_equal(poses, poses2)
    cost, *_ = prob2.evaluate()            # allowed again after solver_end
    assert cost == pytest.approx(s2.final_cost, rel=1e-12)


def test_split_step_timeout_falls_back_to_two_launches(gpu, O, ds, monkeypatch):
    """The single-launch SPLIT steps of the 6x6-block factorisation wait in-kernel for a column's diagonal block.  With the
    wait budget forced to zero (PGO_WAIT_SPINS=0: what a GPU that cannot keep the step's work-groups resident looks
    like) the solve must not fail: the driver repeats the factorisation in the two-launch form and the LM trace is the
    oracle's (same accept / reject sequence, costs 1e-7)."""
    monkeypatch.setenv("PGO_WAIT_SPINS", "0")
    monkeypatch.setenv("PGO_FRONT", "0")
    monkeypatch.setenv("PGO_SFRONT", "0")
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti00.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=8, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, osum, otr = O.solve(og, O.default_options(max_num_iterations=8, linear_solver=0))
    assert s.linear_solver_used == 0 and s.c.factor_kind == 1
    n = min(len(otr), len(s.iterations))
    assert n == len(otr) == len(s.iterations)
    assert list(s.iterations["step_is_successful"][:n]) == [int(v) for v in otr[:n, 8]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)


def test_small_front_single_launch_timeout_falls_back_to_levels(gpu, O, ds, monkeypatch, knobs):
    """The single-launch form of the small-front factorisation (all tree levels in one launch, a front polls its children's
    flags) with the wait budget forced to zero (PGO_WAIT_SPINS=0): the driver repeats the factorisation level by level, keeps
    to that form, and the LM trace is the oracle's; and both forms give bit-identical traces when nothing times out."""
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti00.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    opt = gpu.SolverOptions(max_num_iterations=8, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, osum, otr = O.solve(og, O.default_options(max_num_iterations=8, linear_solver=0))
    traces = []
    for spins, fused in (("0", None), (None, 0), (None, None)):
        monkeypatch.delenv("PGO_WAIT_SPINS", raising=False)
        if spins is not None:
            monkeypatch.setenv("PGO_WAIT_SPINS", spins)
        knobs(factor_fused=fused)
        prob, poses = gpu.problem_from_graph(g)
        s = gpu.solve(opt, prob)
        assert s.linear_solver_used == 0 and s.c.factor_kind == 3
        assert len(otr) == len(s.iterations)
        assert list(s.iterations["step_is_successful"]) == [int(v) for v in otr[:, 8]]
        assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=1e-7)
        traces.append((tuple(float(c) for c in s.iterations["cost"]), poses.tobytes()))
    assert traces[0] == traces[1] == traces[2]


def test_diagonal_information_is_read_as_six_planes_with_identical_results(gpu, ds, knobs):
    """W = diag(1/sigma^2) (the synthetic generators): the kernels read six of the 21 information planes (info_mode 3); the entries
    they skip are exact zeros, so the LM trace and the poses equal those of the 12-plane reads (knob no_diag_info = 1) bit for bit,
    for truncated PCG and for exact steps."""
    g = ds.manhattan_se3(1200, 4200, seed=33)
    out = []
    for ls, kw in ((gpu.BLOCK_JACOBI_PCG, dict(eta=0.1, max_linear_solver_iterations=500)), (gpu.SPARSE_NORMAL_CHOLESKY, {})):
        res = []
        for off in (1, None):
            knobs(no_diag_info=off)
            prob, poses = gpu.problem_from_graph(g)
            s = gpu.solve(gpu.SolverOptions(max_num_iterations=8, linear_solver_type=ls, **kw), prob)
            res.append((tuple(float(c) for c in s.iterations["cost"]), tuple(int(c) for c in s.iterations["linear_solver_iterations"]), poses.tobytes()))
        assert res[0] == res[1]
        out.append(res[0])
    assert len(out) == 2


def test_multifrontal_single_launch_timeout_falls_back_to_launches(gpu, ds, monkeypatch, knobs):
    """The single-launch form of the multifrontal factorisation (every work-group of the round schedule in one grid, stages
    ordered by counters) with the wait budget forced to zero (PGO_WAIT_SPINS=0): the driver repeats the factorisation with one
    launch per phase, keeps to that form, and the trace is bit-identical to the one of either form when nothing times out."""
    g = ds.manhattan_se3(1500, 5000, seed=21)
    opt = gpu.SolverOptions(max_num_iterations=6, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    traces = []
    for spins, fused in (("0", None), (None, 0), (None, 1)):
        monkeypatch.delenv("PGO_WAIT_SPINS", raising=False)
        if spins is not None:
            monkeypatch.setenv("PGO_WAIT_SPINS", spins)
        knobs(factor_fused=fused)
        prob, poses = gpu.problem_from_graph(g)
        s = gpu.solve(opt, prob)
        assert s.linear_solver_used == 0 and s.c.factor_kind == 2
        traces.append((tuple(float(c) for c in s.iterations["cost"]), poses.tobytes()))
    assert traces[0] == traces[1] == traces[2]


@pytest.mark.parametrize("name,exact", [("manhattan1000", True), ("sphere2x20", True), ("manhattan2000", True), ("manhattan1000", False)])
def test_traces_to_convergence_match_oracle(gpu, O, ds, name, exact):
    """The WHOLE trust-region trajectory, not its first iterations: the reference's options (max 300 iterations, default
    tolerances) until the minimizer stops by itself — same number of iterations (40 / 18 / 60 with exact steps, 64 with
    truncated PCG), same accept / reject decision at every one, same stopping reason, costs to 1e-8 relative along the way
    (measured 1e-12 exact; PCG with Ceres' recurrences 4e-11, with the pipelined ones 1.2e-8 — tolerance 1e-7 there), final poses
    to 1e-7 (exact steps) / 1e-5 (truncated PCG: 2e-6 measured — the inexact steps leave the flat directions of the graph to the
    rounding of the CG recurrences)."""
    g = {"manhattan1000": lambda: ds.manhattan_se3(1000, 3500, seed=17),
         "sphere2x20": lambda: ds.sphere_layers(n_spheres=2, rings=20, per_ring=20),
         "manhattan2000": lambda: ds.manhattan_se3(2000, 8000, seed=3)}[name]()
    prob, poses = gpu.problem_from_graph(g)
    ls, cl = (gpu.SPARSE_NORMAL_CHOLESKY, 1) if exact else (gpu.BLOCK_JACOBI_PCG, 2)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=300, linear_solver_type=ls, pcg_cluster_poses=cl), prob)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    # (the oracle runs the CG recurrences the GPU session ran: Summary::cg_form 3 = the pipelined ones of the fused stream)
    oposes, osum, otr = O.solve(og, O.default_options(max_num_iterations=300, linear_solver=0 if exact else 1, pcg_cluster=cl,
                                                      pcg_form=1 if s.cg_form == 3 else 0))
    assert len(s.iterations) == len(otr) and 15 < len(otr) < 300
    assert list(s.iterations["step_is_successful"]) == [int(v) for v in otr[:, 8]]
    # (pipelined recurrences, Summary::cg_form 3: 1.2e-8 measured at iteration 60 of 64 — 3e-9 before the first linearisation of a
    # session took the lean algebra too; exact steps: 1.4e-12 / 1.4e-12 / 4.2e-12 on the three graphs)
    assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=1e-7 if s.cg_form == 3 else 1e-8)
    assert s.termination_type == gpu.CONVERGENCE and s.termination_type == osum.termination_type
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-7 if s.cg_form == 3 else 1e-9)
    # (pipelined recurrences, Summary::cg_form 3: their rounding differs more between two implementations — 5.8e-4 m measured in the
    # flat directions after 64 truncated steps, at costs equal to 1.2e-8)
    assert np.abs(poses - oposes).max() <= (1e-7 if exact else 2e-3 if s.cg_form == 3 else 1e-5)
    if not exact:
        assert s.num_linear_solver_iterations == osum.num_linear_iterations


def test_verbose_switch_and_graph_knob(gpu, ds, monkeypatch, knobs, capfd):
    """PGO_VERBOSE=1: the library's diagnostics on stderr (phase timings of the topology build among them) and nothing else changes;
    the knob graph = 1 (captured hipGraphs of the launch batches instead of eager enqueue): the same LM trace bit for bit."""
    g = ds.manhattan_se3(600, 2000, seed=8)
    opt = dict(max_num_iterations=6, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_form=1)
    monkeypatch.setenv("PGO_NO_PIPELINE", "1")          # (captured batches belong to the host-driven loop)
    prob, p0 = gpu.problem_from_graph(g)
    s0 = gpu.solve(gpu.SolverOptions(**opt), prob)
    capfd.readouterr()
    monkeypatch.setenv("PGO_VERBOSE", "1")
    prob, p1 = gpu.problem_from_graph(g)
    s1 = gpu.solve(gpu.SolverOptions(**opt), prob)
    err = capfd.readouterr().err
    assert "[pgo] prepare:" in err
    monkeypatch.delenv("PGO_VERBOSE")
    knobs(graph=1)
    prob, p2 = gpu.problem_from_graph(g)
    s2 = gpu.solve(gpu.SolverOptions(**opt), prob)
    for s, p in ((s1, p1), (s2, p2)):
        assert np.array_equal(s.iterations["cost"], s0.iterations["cost"]) and np.array_equal(p, p0)
