import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000)
for rep in range(2):
    prob, poses = gpu.problem_from_graph(g)
    t=time.time()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=25, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
    print("rep", rep, len(s.iterations), "%.10e" % s.final_cost, "cg", s.num_linear_solver_iterations, "cg_form", s.cg_form, "wall %.1f ms" % (1e3*(time.time()-t)), flush=True)
