import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
from oracle import oracle as O
k = np.load(os.path.join(ROOT, "tests/golden/kitti00.npz"))
for name, g in [("manhattan300", ds.manhattan_se3(300, 1000, seed=4)), ("kitti00", ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)),
                ("manhattan10k", ds.manhattan_se3(10000, 40000))]:
    prob, poses = pkg.problem_from_graph(g)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    rng = np.random.default_rng(1)
    d2 = rng.uniform(0.1, 1.0, size=g.N * 6); b = rng.normal(size=g.N * 6); b[:6] = 0
    t = time.time(); x, it = prob.linear_solve(d2, b, pkg.SolverOptions(linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)); t1 = time.time() - t
    t = time.time(); x, it = prob.linear_solve(d2, b, pkg.SolverOptions(linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)); t2 = time.time() - t
    xo, _ = O.linear_solve(og, d2, b, linear_solver=0)
    print(name, "linear_solve first %.3fs second %.4fs it=%d err %.2e" % (t1, t2, it, np.abs(x - xo).max() / np.abs(xo).max()))
    t = time.time(); s = pkg.solve(pkg.SolverOptions(max_num_iterations=50, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY), prob); dt = time.time() - t
    t = time.time(); op, osum, otr = O.solve(og, O.default_options(max_num_iterations=50, linear_solver=0)); dto = time.time() - t
    print("  solve: gpu %.3fs (%d its, solver %d, blocks %d levels %d) cost %.9e | oracle %.3fs (%d its, nnzL %d) cost %.9e | max dp %.2e" % (
        dt, s.num_iterations, s.linear_solver_used, s.factor_nnz_blocks, s.factor_levels, s.final_cost, dto, osum.num_iterations, osum.factor_nnz_blocks, osum.final_cost, np.abs(poses[:, :3] - op[:, :3]).max()))
    print("  ", s.message, "| lin %.4f jac %.4f res %.4f total %.4f setup %.4f" % (s.linear_solver_time_in_seconds, s.jacobian_evaluation_time_in_seconds, s.residual_evaluation_time_in_seconds, s.total_time_in_seconds, s.setup_time_in_seconds))
