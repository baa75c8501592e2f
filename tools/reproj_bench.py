"""Batched MotionEstimate solves (pgo_reproj_solve_batch): all candidate pairs of a KITTI-00-sized run at once (20 499 pairs x
300 matched points, synthetic), GPU kernel time against the CPU oracle on a sample.  usage: python tools/reproj_bench.py"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = pgo_loader.load()
INTR = np.array([718.856, 718.856, 607.1928, 185.2157])


def main(n_problems=20499, n_points=300, seed=3):
    rng = np.random.default_rng(seed)
    total = n_problems * n_points
    P = np.c_[rng.uniform(-12, 12, total), rng.uniform(-3, 3, total), rng.uniform(4, 45, total)]
    rv = rng.normal(0, 0.03, (n_problems, 3))
    q_true = Rotation.from_rotvec(rv).as_quat()
    t_true = rng.normal(0, 0.5, (n_problems, 3))
    X = Rotation.from_rotvec(np.repeat(rv, n_points, axis=0)).apply(P) + np.repeat(t_true, n_points, axis=0)
    obs = np.c_[INTR[0] * X[:, 0] / X[:, 2] + INTR[2], INTR[1] * X[:, 1] / X[:, 2] + INTR[3]] + rng.normal(0, 0.5, (total, 2))
    bad = rng.random(total) < 0.05
    obs[bad] += rng.normal(0, 40, (int(bad.sum()), 2))
    ptr = np.arange(n_problems + 1, dtype=np.int64) * n_points
    for mode, qc in (("t only (reference setting)", 1), ("q and t", 0)):
        best = None
        for rep in range(3):
            q = np.ascontiguousarray(q_true if qc else Rotation.from_rotvec(rv + 0.01).as_quat())
            t = np.zeros((n_problems, 3))
            w0 = time.perf_counter()
            summ, ms = pkg.reproj_solve_batch(ptr, P, obs, INTR, q, t, pkg.ReprojOptions(q_constant=qc), return_ms=True)
            wall = time.perf_counter() - w0
            best = ms if best is None else min(best, ms)
        its = summ["num_iterations"]
        err = np.abs(t - t_true).max(axis=1)
        # CPU oracle on a sample
        k = 200
        c0 = time.perf_counter()
        for i in range(k):
            O.reproj_solve(P[ptr[i]:ptr[i + 1]], obs[ptr[i]:ptr[i + 1]], INTR, q_true[i] if qc else Rotation.from_rotvec(rv[i] + 0.01).as_quat(),
                           np.zeros(3), cmask=2 if qc else 0)
        cpu = (time.perf_counter() - c0) / k
        lin_passes = float(summ["num_successful_steps"].sum())        # linearisations (iteration 0 included)
        cost_passes = float((its - 1).clip(min=0).sum())
        flops = n_points * (lin_passes * 330.0 + cost_passes * 70.0)
        byts = n_points * 40.0 * (lin_passes + cost_passes)
        print("%-28s %d problems x %d points: kernel %.2f ms (%.0f problems/ms, wall incl. copies %.1f ms), mean iterations %.2f, "
              "median |t - t_true| %.4f m; ~%.2f TFLOP/s FP64, %.2f TB/s of point reads; CPU oracle %.3f ms/problem -> %.0fx" % (
                  mode, n_problems, n_points, best, n_problems / best, 1e3 * wall, its.mean(), np.median(err),
                  flops / (best * 1e-3) / 1e12, byts / (best * 1e-3) / 1e12, 1e3 * cpu, cpu * n_problems / (best * 1e-3)))


if __name__ == "__main__":
    main()
