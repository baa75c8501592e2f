"""Development aid: the resident stream (pcg_form 3) against the fused one (pcg_form 2): records bit for bit, and ms per LM step at C2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
FIELDS = ["iteration", "step_is_successful", "linear_solver_iterations", "cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius"]
def solve(g, form, cl, its=20):
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cl, pcg_form=form), prob)
    return s, poses
os.environ["PGO_BLOCK"] = "256"
for name, g in (("manhattan 1500", ds.manhattan_se3(1500, 6000, seed=21)), ("manhattan 3001", ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0))):
    for cl in (1, 2):
        a, pa = solve(g, 2, cl); b, pb = solve(g, 3, cl)
        same = all(np.array_equal(a.iterations[f], b.iterations[f]) for f in FIELDS)
        print("%s cluster %d: cg_form %d / %d, its %d / %d, CG %d / %d, records identical %s, poses identical %s, cost %.12e / %.12e, '%s'" % (
            name, cl, a.cg_form, b.cg_form, len(a.iterations), len(b.iterations), a.num_linear_solver_iterations, b.num_linear_solver_iterations,
            same, np.array_equal(pa, pb), a.final_cost, b.final_cost, b.message[:50]), flush=True)
del os.environ["PGO_BLOCK"]
c2 = ds.manhattan_se3(10000, 40000)
for form in (1, 2, 3):
    prob, poses = gpu.problem_from_graph(c2)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=form))
    prob.solver_step(5)
    cg_us = 1e3 * prob.time_kernel("uni_cg", 3)
    ts = []
    for rep in range(5):
        prob.solver_reset(); prob.solver_step(5)
        t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
    s = prob.solver_end()
    print("C2 form %d (cg_form %d): CG iteration in situ %.2f us; LM step %.4f ms (median of 5); final cost %.10e" % (form, s.cg_form, cg_us, 1e3 * float(np.median(ts)), s.final_cost), flush=True)
