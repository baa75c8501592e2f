import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1001, 3700)
g = ds.manhattan_se3(n, e, seed=31)
res = []
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
    prob, poses = pkg.problem_from_graph(g)
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=8, linear_solver_type=pkg.BLOCK_JACOBI_PCG), prob)
    res.append((tuple(int(x) for x in s.iterations["linear_solver_iterations"]), repr(s.final_cost)))
for r in res: print(r)
print("deterministic:", len(set(res)) == 1)
