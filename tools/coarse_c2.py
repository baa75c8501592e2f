"""BASELINE configs[1] with the PCG's coarse level (options.pcg_coarse_aggregate) on the GPU: every policy from dead reckoning to its own
stop — final cost against the exact path's, LM / CG iterations, wall time (pgo_solve, host buffers in and out).
usage (GPU box): python tools/coarse_c2.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000, seed=20260928)


def run(**kw):
    best = None
    for _ in range(2):
        prob, poses = gpu.problem_from_graph(g)
        t = time.perf_counter()
        s = gpu.solve(gpu.SolverOptions(**kw), prob)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]:
            best = (dt, s, poses)
    return best


dt, ex, pe = run(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
print("exact steps                      : cost %.6e, %4d LM iterations, %.3f s" % (ex.final_cost, ex.num_iterations - 1, dt), flush=True)
for name, kw in (("cluster Jacobi, eta 0.1", dict(eta=0.1)), ("cluster Jacobi, eta 1e-5", dict(eta=1e-5, max_linear_solver_iterations=3000)),
                 ("+ coarse level agg 32, eta 0.1", dict(eta=0.1, pcg_coarse_aggregate=32)), ("+ coarse level agg 64, eta 0.1", dict(eta=0.1, pcg_coarse_aggregate=64)),
                 ("+ coarse level agg 128, eta 0.1", dict(eta=0.1, pcg_coarse_aggregate=128))):
    dt, s, p = run(max_num_iterations=3000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, **kw)
    c = s.iterations["cost"]
    hit = np.nonzero(c <= ex.final_cost * (1 + 1e-3))[0]
    t_hit = None if len(hit) == 0 else s.total_time_in_seconds * hit[0] / max(1, len(c) - 1)
    print("%-33s: cost %.6e (%+.2f %% vs exact), %4d LM / %6d CG iterations, %.3f s wall (%.3f s in the LM loop, %.3f ms per LM iteration)%s" % (
        name, s.final_cost, 100 * (s.final_cost / ex.final_cost - 1), s.num_iterations - 1, s.num_linear_solver_iterations, dt, s.total_time_in_seconds,
        1e3 * s.total_time_in_seconds / max(1, s.num_iterations - 1), "" if t_hit is None else "; exact cost x (1 + 1e-3) reached after %.3f s" % t_hit), flush=True)
