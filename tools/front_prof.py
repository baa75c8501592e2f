"""Profiling driver: N factorisations + solves of one configuration (for rocprofv3 --kernel-trace --stats).
usage: python tools/front_prof.py c2|c5|m2000 [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
os.environ.setdefault("PGO_FRONT", "1")
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = {"c2": ds.manhattan_se3, "c5": ds.sphere_layers, "m2000": lambda: ds.manhattan_se3(2000, 8000, seed=3)}[which]()
prob, poses = gpu.problem_from_graph(g)
prob.solver_begin(gpu.SolverOptions(max_num_iterations=4, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
tf = prob.time_kernel("front_factor", reps)
ts = prob.time_kernel("front_solve", reps)
print("%s factor %.3f ms solve %.3f ms" % (which, tf, ts))
prob.solver_end()
