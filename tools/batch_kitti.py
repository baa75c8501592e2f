"""Throughput of pgo_solve_batch on KITTI-00-scale graphs with the reference's options: n copies of the KITTI-00 replay graph
as one batch (one launch sequence) vs one pgo_solve.  usage: python tools/batch_kitti.py [n ...]"""
import os
import sys
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
    walls = []
    for rep in range(6):
        prob, _ = pkg.problem_from_graph(g)
        t0 = time.perf_counter()
        s1 = pkg.solve(opt, prob)
        walls.append(time.perf_counter() - t0)
    single = sorted(walls)[len(walls) // 2]
    its = len(s1.iterations) - 1
    print("single pgo_solve: %.2f ms wall (median of 6), %d iterations, %.0f LM it/s, final %.9e" % (1e3 * single, its, its / single, s1.final_cost), flush=True)
    for n in counts:
        walls = []
        for rep in range(4):
            pairs = [pkg.problem_from_graph(g) for _ in range(n)]
            t0 = time.perf_counter()
            sums = pkg.solve_batch(opt, [p for p, _ in pairs])
            walls.append(time.perf_counter() - t0)
        w = sorted(walls)[len(walls) // 2]
        assert all(len(s.iterations) - 1 == its for s in sums) and all(abs(s.final_cost - s1.final_cost) < 1e-7 for s in sums)
        print("batch of %3d: %.2f ms wall (setup %.2f), %.3f ms per graph, %.0f LM it/s aggregate = %.1fx one-at-a-time" % (
            n, 1e3 * w, 1e3 * sums[0].c.setup_time_in_seconds, 1e3 * w / n, n * its / w, single * n / w), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64])
