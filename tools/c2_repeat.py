import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000)
for rep in range(3):
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    it = s.iterations
    print("rep", rep, len(it), "%.10e" % s.final_cost, "cost[60] %.15e cost[100] %.15e" % (it["cost"][60], it["cost"][100]), flush=True)
