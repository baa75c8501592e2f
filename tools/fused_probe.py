"""Development aid: the fused universal stream (pcg_form 2) against the two-kernel stream (pcg_form 1) and the oracle's two CG forms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
from oracle import oracle as O

def run(g, form, cluster, its, eta=0.1):
    prob, poses = gpu.problem_from_graph(g)
    t = time.time()
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster, pcg_form=form, eta=eta), prob)
    return s, poses, time.time() - t

small = ds.manhattan_se3(2000, 8000, seed=5)
og = O.Graph(small.poses, small.ia, small.ib, small.meas, small.sqrt_info)
for cl in (1, 2):
    for form in (1, 2):
        s, p, dt = run(small, form, cl, 25)
        op, osum, otr = O.solve(og, O.default_options(max_num_iterations=25, linear_solver=1, pcg_cluster=cl, pcg_form=form - 1))
        n = min(len(otr), len(s.iterations))
        same_dec = list(s.iterations["step_is_successful"][:n]) == [int(x) for x in otr[:n, 8]]
        same_cg = list(s.iterations["linear_solver_iterations"][:n]) == [int(x) for x in otr[:n, 7]]
        print("small cl %d form %d: cg_form %d its %d cg %d cost %.10e | oracle cg %d cost %.10e | decisions %s cg counts %s maxrel cost %.2e term '%s'" % (
            cl, form, s.cg_form, len(s.iterations), s.num_linear_solver_iterations, s.final_cost, osum.num_linear_iterations, osum.final_cost,
            same_dec, same_cg, np.abs(s.iterations["cost"][:n] / otr[:n, 1] - 1).max(), s.message[:60]), flush=True)
        if not same_cg:
            print("   gpu", list(s.iterations["linear_solver_iterations"][:n])); print("   ora", [int(x) for x in otr[:n, 7]])
c2 = ds.manhattan_se3(10000, 40000)
for form in (1, 2, 2, 1):
    s, p, dt = run(c2, form, 2, 25)
    print("C2 form %d: cg_form %d its %d cg %d cost %.10e wall %.1f ms" % (form, s.cg_form, len(s.iterations), s.num_linear_solver_iterations, s.final_cost, 1e3 * dt), flush=True)
# device-resident stepping as bench.py times it
for form in (1, 2):
    prob, poses = gpu.problem_from_graph(c2)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=form))
    prob.solver_step(5)
    ts = []
    for rep in range(5):
        prob.solver_reset(); prob.solver_step(5)
        t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
    s = prob.solver_end()
    print("C2 stepping form %d (cg_form %d): ms per LM step %s" % (form, s.cg_form, ["%.4f" % (1e3 * x) for x in ts]), flush=True)
    if form == 2:
        prob, poses = gpu.problem_from_graph(c2)
        prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=2))
        prob.trace_start(20000)
        prob.solver_step(25)
        rec, hl, hs = prob.trace_read()
        prob.solver_end()
        names = {0: "idle", 1: "head", 2: "w0", 3: "cg", 4: "tail", 5: "lin"}
        dur = (rec[:, 2] - rec[:, 1]) / 100.0
        gap = (rec[1:, 1] - rec[:-1, 2]) / 100.0
        for op in range(6):
            m = rec[:, 0] == op
            if m.any(): print("  %-5s launches %4d  avg %.2f us  median %.2f" % (names[op], m.sum(), dur[m].mean(), np.median(dur[m])))
        print("  gaps: avg %.2f us median %.2f; host: %d launches, %.2f us per launch; span %.1f us" % (gap.mean(), np.median(gap), hl, 1e6 * hs / max(hl, 1), (rec[-1, 2] - rec[0, 1]) / 100.0))
        print("  tick " + str(int(rec[0, 1])))
        print("  cg_time_kernel us:", 1e3 * 0)
