"""r06 experiment, ORACLE ONLY (CPU): does an aggregation coarse level make the truncated PCG (Ceres' default forcing term 0.1, what the
headline steps use) reach the exact path's answer on BASELINE configs[1]?  Two-level additive preconditioner: 2-pose cluster Jacobi +
aggregates of `agg` consecutive poses with six rigid-body modes each, Galerkin coarse matrix (oracle/pgo_oracle.cpp pcg_solve,
pcg_cluster = -agg).  Every policy runs to its own stop (function tolerance 1e-6, <= 1000 LM iterations) from dead reckoning.
usage: python tools/two_level_oracle.py [poses edges]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402
from oracle import oracle as O  # noqa: E402

ds = pgo_loader.datasets()
n, e = (int(a) for a in (sys.argv[1:3] + ["10000", "40000"][len(sys.argv) - 1:]))
g = ds.manhattan_se3(n, e, seed=20260928)
og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
nthr = min(os.cpu_count() or 1, 32)
t = time.time()
pe, se, te = O.solve(og, O.default_options(max_num_iterations=1000, linear_solver=0, num_threads=nthr))
print("exact steps: cost %.6e after %d LM iterations (%.1f s)" % (se.final_cost, se.num_iterations - 1, time.time() - t), flush=True)
for eta in (0.1, 0.01):
    for cl in (2, -32, -64, -128, -256):
        t = time.time()
        p, s, tr = O.solve(og, O.default_options(max_num_iterations=1000, linear_solver=1, pcg_cluster=cl, pcg_form=1, eta=eta,
                                                 max_linear_solver_iterations=500, num_threads=nthr))
        c25 = tr[min(25, len(tr) - 1), 1]
        print("eta %-5g %-28s final %.6e (%+.2f %% vs exact) after %4d LM / %6d CG iterations; cost after 25 LM iterations %.6e; max |dp| to exact %.1f m (%.0f s)" % (
            eta, "2-pose Jacobi" if cl == 2 else "+ coarse level, agg %d" % -cl, s.final_cost, 100 * (s.final_cost / se.final_cost - 1),
            s.num_iterations - 1, s.num_linear_iterations, c25, float(np.linalg.norm(p[:, :3] - pe[:, :3], axis=1).max()), time.time() - t), flush=True)
