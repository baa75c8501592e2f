"""Development aid: launch trace of the resident stream at C2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
if os.environ.get("PGO_AB_LIB"): gpu.LIB_PATH = os.environ["PGO_AB_LIB"]      # development: trace another build of the library
c2 = ds.manhattan_se3(10000, 40000)
prob, poses = gpu.problem_from_graph(c2)
prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=3))
prob.solver_step(5)
ts = []
for rep in range(5):
    prob.solver_reset(); prob.solver_step(5)
    t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
print("untraced %.4f ms per LM step (median of 5 x 20 steps)" % (1e3 * float(np.median(ts))))
prob.solver_reset(); prob.solver_step(5)
prob.trace_start(4000)
t = time.perf_counter(); ran, done = prob.solver_step(20); wall = time.perf_counter() - t
rec, hl, hs = prob.trace_read()
s = prob.solver_end()
print("cg_form", s.cg_form, "traced %.4f ms per step" % (1e3 * wall / ran))
names = {0: "idle", 1: "head", 3: "cg", 4: "tail", 5: "lin"}
dur = (rec[:, 2] - rec[:, 1]) / 100.0
gap = np.append((rec[1:, 1] - rec[:-1, 2]) / 100.0, 0)
for op, nm in names.items():
    m = rec[:, 0] == op
    if m.any(): print("  %-5s %3d launches: kernel %.2f us, gap behind %.2f us" % (nm, m.sum(), dur[m].mean(), gap[m].mean()))
cg = rec[rec[:, 0] == 3]
ph = np.array([[(int(w) >> (16 * k)) & 0xffff for k in range(4)] for w in cg[:, 3]], dtype=float)
its = ph[:, 3] + 2          # barriers per launch: w0 + iterations ... (cnt + 1 turns of the loop)
print("  cg launches: mean CG iterations %.1f; per turn of the loop (work-group 0): fold + product + recurrences + publish %.2f us, grid barrier %.2f us, requests behind it %.2f us" % (
    ph[:, 3].mean(), (ph[:, 0] / (ph[:, 3] + 1)).mean() / 100, (ph[:, 1] / (ph[:, 3] + 1)).mean() / 100, (ph[:, 2] / (ph[:, 3] + 1)).mean() / 100))
for op, nm, labels in ((1, "head", ("accept-finish done", "its chunk done (damping, Jacobi block, b, M^-1 b)", "known to be last", "end")),
                       (4, "tail", ("loops done", "known to be last", "partials folded", "decided"))):
    w = rec[rec[:, 0] == op][:, 3]
    w = w[w != 0]
    if len(w):
        st = np.array([[(int(x) >> (16 * k)) & 0xffff for k in range(4)] for x in w], dtype=float) / 100.0
        print("  %s, the last work-group to arrive (us from its top, median of %d): %s" % (nm, len(w), ", ".join("%s %.2f" % (l, v) for l, v in zip(labels, np.median(st, axis=0)))))
