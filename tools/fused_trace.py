"""Development aid: launch trace of the fused universal stream at C2 (pgo_solver_trace_*), 25 LM steps.  [block]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
if len(sys.argv) > 1: os.environ["PGO_BLOCK"] = sys.argv[1]
c2 = ds.manhattan_se3(10000, 40000)
prob, poses = gpu.problem_from_graph(c2)
prob.solver_begin(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=2))
prob.solver_step(5)
ts = []
for rep in range(3):
    prob.solver_reset(); prob.solver_step(5)
    t = time.perf_counter(); ran, done = prob.solver_step(20); ts.append((time.perf_counter() - t) / max(ran, 1))
print("untraced: %.4f ms per LM step" % (1e3 * np.median(ts)))
prob.solver_reset(); prob.solver_step(5)
prob.trace_start(20000)
t = time.perf_counter(); ran, done = prob.solver_step(20); wall = time.perf_counter() - t
rec, hl, hs = prob.trace_read()
prob.solver_end()
names = {0: "idle", 1: "head", 2: "w0", 3: "cg", 4: "tail", 5: "lin"}
live = rec[:, 0] > 0
last = np.nonzero(live)[0].max()
idle = rec[last + 1:]
if len(idle) > 2:
    print("  idle  %4d launches behind the pause: top-to-last-end %.2f us, period %.2f us" % (len(idle), ((idle[:, 2] - idle[:, 1]) / 100.0).mean(), (np.diff(idle[:, 1]) / 100.0).mean()))
rec = rec[: last + 1]
dur = (rec[:, 2] - rec[:, 1]) / 100.0
period = np.diff(rec[:, 1]) / 100.0
gap = (rec[1:, 1] - rec[:-1, 2]) / 100.0
print("traced: %.4f ms per LM step (host wall), %d launches, device span %.1f us" % (1e3 * wall / ran, len(rec), (rec[-1, 2] - rec[0, 1]) / 100.0))
for op in range(6):
    m = rec[:-1, 0] == op
    if m.any():
        print("  %-5s %4d launches: top-to-last-end %.2f us (median %.2f), gap behind it %.2f us (median %.2f), period %.2f us; per LM step %.1f us" % (
            names[op], m.sum(), dur[:-1][m].mean(), np.median(dur[:-1][m]), gap[m].mean(), np.median(gap[m]), period[m].mean(), period[m].sum() / ran))
print("  host: %d launches enqueued, %.2f us per launch" % (hl, 1e6 * hs / max(hl, 1)))

def phases(w):
    return [int((w >> (16 * k)) & 0xffff) / 100.0 for k in range(4)]
for op, label in ((3, "cg  (wg 0: product, fold, rows, end)"), (4, "tail (deciding wg: loops, last known, folded, decided)")):
    rows = np.array([phases(int(w)) for w in rec[rec[:, 0] == op][5:, 3]])
    if len(rows):
        print("  phases %s: median us %s" % (label, np.round(np.median(rows, axis=0), 2)))
