"""Development aid: device-decided vs host-decided LM records of one solve, field by field."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(1000, 3500, seed=17)
def run(host, **kw):
    os.environ["PGO_NO_PIPELINE"] = "1" if host else "0"
    prob, poses = gpu.problem_from_graph(g)
    return gpu.solve(gpu.SolverOptions(max_num_iterations=120, **kw), prob), poses
for kw in (dict(linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), dict(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)):
    a, pa = run(False, **kw); b, pb = run(True, **kw)
    print(kw, len(a.iterations), len(b.iterations), a.message, "|", b.message)
    n = min(len(a.iterations), len(b.iterations))
    for f in a.iterations.dtype.names:
        x, y = a.iterations[f][:n], b.iterations[f][:n]
        bad = np.nonzero(x != y)[0]
        if len(bad):
            k = bad[0]
            print("  field", f, "differs at", bad[:8], "first: %r vs %r" % (x[k], y[k]))
    print("  poses equal:", np.array_equal(pa, pb), "times:", a.total_time_in_seconds, b.total_time_in_seconds, a.linear_solver_time_in_seconds, a.jacobian_evaluation_time_in_seconds)
