"""Solution quality of the truncated-PCG policy against the exact-step path (the reference's SPARSE_NORMAL_CHOLESKY setting):
final cost of each at its own stop for a sweep of the forcing term eta, distance between the solutions, and the time each needs
to reach the exact path's final cost within 1e-2 (the exact path's own run-to-run spread on C2 is 3e-3: DESIGN.md section 5).   usage (GPU box): python tools/pcg_quality.py [c2|c4]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
g = ds.manhattan_se3(10000, 40000) if which == "c2" else ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)


def solve(**kw):
    prob, poses = gpu.problem_from_graph(g)
    t = time.perf_counter()
    s = gpu.solve(gpu.SolverOptions(**kw), prob)
    return s, poses, time.perf_counter() - t


def time_to(s, target):
    """seconds (solver clock, prorated over the iteration records) until the cost first drops to `target`"""
    c = s.iterations["cost"]
    hit = np.nonzero(c <= target)[0]
    if len(hit) == 0:
        return None
    return s.total_time_in_seconds * (hit[0] / max(1, len(c) - 1))


ref = None
if which == "c2":
    ref, pref, wall = solve(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    print("exact: %d its, final %.6e, %.1f ms" % (ref.num_iterations, ref.final_cost, 1e3 * wall), flush=True)
for eta in (0.1, 1e-2, 1e-3, 1e-4, 1e-5):
    s, p, wall = solve(max_num_iterations=3000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=eta,
                       max_linear_solver_iterations=3000)
    line = "pcg eta %g: %d its, %d CG, final %.6e, %.1f ms, %s" % (eta, s.num_iterations, s.num_linear_solver_iterations, s.final_cost, 1e3 * wall, s.message[:28])
    if ref is not None:
        tgt = ref.final_cost * (1 + 1e-2)
        line += " | rel to exact %+.2e, max |dp| %.3f m, time to exact*(1+1e-2): pcg %s, exact %s" % (
            s.final_cost / ref.final_cost - 1, np.linalg.norm(p[:, :3] - pref[:, :3], axis=1).max(), time_to(s, tgt), time_to(ref, tgt))
    print(line, flush=True)
