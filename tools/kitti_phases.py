"""Development aid: phase timings (PGO_VERBOSE) and repeated wall times of the KITTI-00 replay solve with the reference's options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
k = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kitti00.npz"))
g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
for rep in range(6):
    if rep == 5: os.environ["PGO_VERBOSE"] = "1"
    t0 = time.perf_counter()
    prob, poses = pkg.problem_from_graph(g)
    t1 = time.perf_counter()
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY), prob)
    t2 = time.perf_counter()
    print("rep %d: problem %.2f ms, solve %.2f ms (setup %.2f), %d iterations, final %.6e" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * s.c.setup_time_in_seconds, len(s.iterations) - 1, s.final_cost), flush=True)
    del prob
