"""Exact-solver behaviour on mesh-like graphs (sphere layers, BASELINE config 5): GPU SPARSE_NORMAL_CHOLESKY request vs the
CPU oracle's exact solve.  usage: python tools/sphere_check.py [n_spheres ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
from oracle import oracle as O
for ns in [int(a) for a in sys.argv[1:]] or [1, 10]:
    g = ds.sphere_layers(n_spheres=ns)
    prob, poses = pkg.problem_from_graph(g)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    for lst, name in [(pkg.SPARSE_NORMAL_CHOLESKY, "exact"), (pkg.BLOCK_JACOBI_PCG, "pcg")]:
        prob2, poses2 = pkg.problem_from_graph(g)
        t = time.time(); s = pkg.solve(pkg.SolverOptions(max_num_iterations=30, linear_solver_type=lst, pcg_cluster_poses=2), prob2); dt = time.time() - t
        print("spheres %d N %d E %d %s: gpu %.3fs its %d solver_used %d blocks %d levels %d cg %d cost %.9e lin %.4f total %.4f setup %.4f" % (
            ns, g.N, g.E, name, dt, s.num_iterations, s.linear_solver_used, s.factor_nnz_blocks, s.factor_levels, s.num_linear_iterations if hasattr(s, "num_linear_iterations") else -1,
            s.final_cost, s.linear_solver_time_in_seconds, s.total_time_in_seconds, s.setup_time_in_seconds), flush=True)
    t = time.time(); op, osum, otr = O.solve(og, O.default_options(max_num_iterations=30, linear_solver=0)); dto = time.time() - t
    print("   oracle exact: %.3fs its %d nnzL %d flops %.3e cost %.9e lin %.3f" % (dto, osum.num_iterations, osum.factor_nnz_blocks, osum.factor_flops, osum.final_cost, osum.linear_solver_seconds), flush=True)
