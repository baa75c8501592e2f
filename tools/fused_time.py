"""Development aid: in-situ time of one CG iteration (pgo_time_kernel 'uni_cg') and ms per LM step of the two universal streams at C2,
for the work-group sizes given on the command line (default 128 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
if os.environ.get("PGO_AB_LIB"): gpu.LIB_PATH = os.environ["PGO_AB_LIB"]      # development: time another build of the library
c2 = ds.manhattan_se3(10000, 40000)
blocks = [int(a) for a in sys.argv[1:]] or [128, 256]
for B in blocks:
    os.environ["PGO_BLOCK"] = str(B)
    for form in (1, 2):
        prob, poses = gpu.problem_from_graph(c2)
        prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=form))
        prob.solver_step(5)
        cg_us = 1e3 * prob.time_kernel("uni_cg", 3)
        ts = []
        for rep in range(5):
            prob.solver_reset(); prob.solver_step(5)
            t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
        s = prob.solver_end()
        cg_per_step = s.num_linear_solver_iterations / max(1, len(s.iterations) - 1)
        print("block %d form %d (cg_form %d): CG iteration in situ %.2f us; LM step %.4f ms (median of 5); %.1f CG per step" % (
            B, form, s.cg_form, cg_us, 1e3 * float(np.median(ts)), cg_per_step), flush=True)
