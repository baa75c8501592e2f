"""Where does the GPU's exact-step LM trajectory on BASELINE configs[1] (Manhattan 10 k / 40 k, reference options) part from the
oracle's (tests/golden/c2_exact_trace.npz)?  Prints, iteration by iteration, the relative differences of cost / radius /
relative decrease and the first iteration at which a decision differs.  Run on the GPU box: python tools/c2_exact_divergence.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
fx = np.load(os.path.join(ROOT, "tests", "golden", "c2_exact_trace.npz"))
tr = fx["trace"]
g = ds.manhattan_se3(10000, 40000, seed=20260928)
assert int(np.asarray(g.ia, dtype=np.int64).sum()) == int(fx["checksum_ia"])
prob, poses = gpu.problem_from_graph(g)
s = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
it = s.iterations
print("gpu: %d records, final cost %.9e, %s | oracle: %d records, final cost %.9e" % (
    len(it), s.final_cost, s.message, len(tr), float(fx["final_cost"])))
n = min(len(it), len(tr))
first = None
for k in range(n):
    dc = abs(it["cost"][k] - tr[k, 1]) / abs(tr[k, 1])
    dr = abs(it["trust_region_radius"][k] - tr[k, 6]) / abs(tr[k, 6])
    drho = abs(it["relative_decrease"][k] - tr[k, 5])
    same = int(it["step_is_successful"][k]) == int(tr[k, 8])
    if k < 12 or k % 10 == 0 or not same or dc > 1e-9:
        print("it %3d  cost %.12e / %.12e (rel %.1e)  radius rel %.1e  rho %.6f / %.6f (abs %.1e)  ok %d/%d" % (
            k, it["cost"][k], tr[k, 1], dc, dr, it["relative_decrease"][k], tr[k, 5], drho, it["step_is_successful"][k], int(tr[k, 8])))
    if first is None and (not same or dc > 1e-6):
        first = k
        print("  ^^^ first departure at iteration", k)
        if k + 3 < n:
            continue
print("first departure:", first)
