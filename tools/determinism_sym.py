"""Development aid: a symmetric-form PCG session (k_linearize_lean, k_spmv_sym) repeated four times on one GPU — CG iteration counts,
final cost and the sum of |pose entries| must agree to the last bit.   usage (GPU box): python tools/determinism_sym.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGO_SYM"] = "1"; os.environ["PGO_NO_PIPELINE"] = "1"
import numpy as np, pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(20000, 150000, seed=7, loop_radius=3.0)
res = []
for rep in range(4):
    prob, poses = pkg.problem_from_graph(g)
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=80, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
    res.append((tuple(int(x) for x in s.iterations["linear_solver_iterations"]), repr(s.final_cost), float(np.abs(poses).sum()).hex()))
for r in res: print(r[1], r[2], sum(r[0]), len(r[0]))
print("deterministic:", len(set(res)) == 1)
