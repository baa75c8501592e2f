"""Race hunt, second kind: the same solve on several host threads / HIP streams at once (kernels of different problems share the
CUs, waves of one workgroup drift apart, device-pool blocks change hands).  Every thread must reproduce the solo run bit for bit.
usage: python tools/concurrent_stress.py [threads] [repeats]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
EX, PCG = pkg.SPARSE_NORMAL_CHOLESKY, pkg.BLOCK_JACOBI_PCG
cases = [("kitti00 exact", ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None), dict(max_num_iterations=100, linear_solver_type=EX)),
         ("manhattan 3000/9000 exact", ds.manhattan_se3(3000, 9000, seed=4), dict(max_num_iterations=12, linear_solver_type=EX)),
         ("sphere 3x16x16 exact", ds.sphere_layers(n_spheres=3, rings=16, per_ring=16), dict(max_num_iterations=12, linear_solver_type=EX)),
         ("manhattan 5000/20000 pcg", ds.manhattan_se3(5000, 20000, seed=6), dict(max_num_iterations=15, linear_solver_type=PCG, eta=0.1, max_linear_solver_iterations=500)),
         ("manhattan 800/2400 pcg huber", ds.manhattan_se3(800, 2400, seed=8, outlier_fraction=0.05) if "outlier_fraction" in ds.manhattan_se3.__code__.co_varnames else ds.manhattan_se3(800, 2400, seed=8),
          dict(max_num_iterations=15, linear_solver_type=PCG, eta=0.1, max_linear_solver_iterations=500))]


def run(g, kw):
    prob, poses = pkg.problem_from_graph(g)
    s = pkg.solve(pkg.SolverOptions(**kw), prob)
    return (tuple(float(c) for c in s.iterations["cost"]), tuple(int(c) for c in s.iterations["linear_solver_iterations"]), poses.tobytes(), s.c.factor_kind)


bad = 0
for name, g, kw in cases:
    solo = run(g, kw)
    for rep in range(repeats):
        out = [None] * threads
        bar = threading.Barrier(threads)

        def work(i):
            bar.wait()
            out[i] = run(g, kw)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        n_diff = sum(o != solo for o in out)
        bad += n_diff > 0
        print("%-30s rep %d: kind %d, %3d iterations, %d of %d threads differ from the solo run %s" % (
            name, rep, solo[3], len(solo[0]) - 1, n_diff, threads, "<-- MISMATCH" if n_diff else ""), flush=True)
print("mismatching rounds:", bad)
