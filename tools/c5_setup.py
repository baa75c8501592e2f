"""Development aid: where the set-up of an exact request on BASELINE configs[4]'s graph goes (PGO_VERBOSE phases), twice in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
g = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931) if which == "c5" else ds.manhattan_se3(10000, 40000)
for k in range(2):
    prob, poses = gpu.problem_from_graph(g)
    t = time.perf_counter()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=3, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    print("solve %d: wall %.1f ms, setup %.1f ms, total %.1f ms, %d iterations" % (k, 1e3 * (time.perf_counter() - t), 1e3 * s.setup_time_in_seconds, 1e3 * s.total_time_in_seconds, s.num_iterations), flush=True)
