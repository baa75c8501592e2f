"""Development aid: only batched KITTI-00 solves (for rocprofv3 --kernel-trace).  usage: batch_profile.py [n] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
opt = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
for _ in range(reps):
    pairs = [pkg.problem_from_graph(g) for _ in range(n)]
    sums = pkg.solve_batch(opt, [p for p, _ in pairs])
print(n, reps, sums[0].final_cost, len(sums[0].iterations))
