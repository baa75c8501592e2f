"""Development aid: CG iterations per LM step of the truncated PCG (eta = 0.1) at BASELINE configs[1] for 6 x 6, 12 x 12 and 24 x 24 Jacobi blocks
(1, 2, 4 poses per cluster), first 25 LM iterations from dead reckoning, and what the two-kernel stream takes per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
c2 = ds.manhattan_se3(10000, 40000, seed=20260928)
for cl in (1, 2, 4):
    prob, poses = gpu.problem_from_graph(c2)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cl, pcg_form=1))
    prob.solver_step(5)
    ts = []
    for rep in range(3):
        prob.solver_reset(); prob.solver_step(5)
        t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
    s = prob.solver_end()
    it = np.asarray(s.iterations["linear_solver_iterations"][1:26], dtype=int)
    print("cluster %d: %.2f CG per LM step over the first 25 (%s), cost after 25: %.6e, LM step %.4f ms (two-kernel stream)" % (
        cl, it.mean(), " ".join(str(v) for v in it), s.iterations["cost"][min(25, len(s.iterations) - 1)], 1e3 * float(np.median(ts))), flush=True)
