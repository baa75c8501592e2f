"""Aggregate throughput of several INDEPENDENT KITTI-00-scale solves with the reference's options (exact steps) running
concurrently on ONE GPU: one host thread and one HIP stream per problem.  usage: python tools/concurrent_kitti.py [n ...]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))


def main(counts):
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    opt = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
    pkg.solve(opt, pkg.problem_from_graph(g)[0])      # warm-up
    for n in counts:
        best = None
        for rep in range(3):
            probs = [pkg.problem_from_graph(g) for _ in range(n)]
            res = [None] * n
            bar = threading.Barrier(n + 1)

            def work(i):
                bar.wait()
                res[i] = pkg.solve(opt, probs[i][0])
                bar.wait()

            ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in ths:
                t.start()
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            dt = time.perf_counter() - t0
            for t in ths:
                t.join()
            assert all(abs(r.final_cost - res[0].final_cost) < 1e-9 for r in res)
            best = dt if best is None else min(best, dt)
        its = len(res[0].iterations) - 1
        print("%2d concurrent KITTI-00 solves: %.2f ms wall for all, %.2f ms per solve, %.0f LM it/s aggregate (final %.6e)" % (
            n, 1e3 * best, 1e3 * best / n, n * its / best, res[0].final_cost), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16])
