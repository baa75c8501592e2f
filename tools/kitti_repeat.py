"""Run-to-run determinism of the exact KITTI-00 solve: n solves of the same graph (a new problem each time) must give the same
iteration count and bit-identical final cost.  usage: python tools/kitti_repeat.py [n] (env switches apply, e.g. PGO_NO_SPECULATION=1)"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
opt = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seen = collections.Counter()
first_diff = None
ref = None
for rep in range(n):
    prob, _ = pkg.problem_from_graph(g)
    s = pkg.solve(opt, prob)
    costs = tuple(float(c) for c in s.iterations["cost"])
    key = (len(s.iterations) - 1, repr(s.final_cost))
    seen[key] += 1
    if ref is None:
        ref = costs
    elif costs != ref and first_diff is None:
        m = min(len(costs), len(ref))
        i = next((i for i in range(m) if costs[i] != ref[i]), m)
        first_diff = (rep, i, ref[i] if i < len(ref) else None, costs[i] if i < len(costs) else None)
for key, cnt in seen.most_common():
    print(cnt, key)
print("first difference (rep, iteration, reference cost, cost):", first_diff)
print("deterministic:", len(seen) == 1)
