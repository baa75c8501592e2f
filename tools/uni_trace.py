"""Development aid for the universal LM stream: (1) python tools/uni_trace.py run [steps] — C2 (Manhattan 10 k / 40 k) PCG, `steps`
LM iterations through pgo_solver_step, to be wrapped in rocprofv3 --kernel-trace; (2) python tools/uni_trace.py show <results.db>
— the launches of the stream in order: kernel, duration, gap before it (the operation of a launch shows in its duration)."""
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps):
    import pgo_loader
    gpu = pgo_loader.load()
    ds = pgo_loader.datasets()
    g = ds.manhattan_se3(10000, 40000, seed=20260928)
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    prob.solver_begin(opt)
    prob.solver_step(steps)
    prob.solver_reset()
    import time
    t0 = time.perf_counter()
    ran, done = prob.solver_step(steps)
    dt = time.perf_counter() - t0
    s = prob.solver_end()
    print("steps %d ran %d: %.3f ms per step, cg %d" % (steps, ran, 1e3 * dt / max(ran, 1), s.num_linear_solver_iterations))


def show(db, limit=400):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else "start_timestamp"
    ec = "end" if "end" in cols else "end_timestamp"
    rows = c.execute("select name, %s, %s from kernels order by %s" % (sc, ec, sc)).fetchall()
    # the second run (after the reset): from the last k_lm_budget on
    idx = [i for i, r in enumerate(rows) if "k_lm_budget" in r[0]]
    a = idx[-1] if idx else 0
    prev = rows[a][2]
    agg = {}
    for i, (n, s, e) in enumerate(rows[a:a + int(limit)]):
        m = re.search(r"(k_[a-z_]+)", n)
        nm = m.group(1) if m else n[:30]
        d, gap = (e - s) / 1e3, (s - prev) / 1e3
        print("%4d %-14s dur %7.2f gap %6.2f" % (i, nm, d, gap))
        prev = e
    tot = {}
    for n, s, e in rows[a:]:
        m = re.search(r"(k_[a-z_]+)", n)
        nm = m.group(1) if m else n[:30]
        t = tot.setdefault(nm, [0, 0.0])
        t[0] += 1
        t[1] += (e - s) / 1e3
    span = (rows[-1][2] - rows[a][1]) / 1e3
    print("span %.1f us" % span)
    for nm, (cnt, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-16s x%5d %9.1f us  avg %.2f" % (nm, cnt, t, t / cnt))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    else:
        show(*sys.argv[2:])
