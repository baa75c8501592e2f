"""Development check of the multifrontal solver on the GPU: exact linear solves vs the oracle on small graphs, timing of
factor / solve on the BASELINE configurations.  usage: python tools/front_gpu_check.py [quick|full]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
from oracle import oracle as O  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
os.environ["PGO_FRONT"] = "1"


def check(name, g, seed=1):
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    prob, poses = gpu.problem_from_graph(g)
    rng = np.random.default_rng(seed)
    d2 = rng.uniform(0.1, 1.0, size=g.N * 6)
    b = rng.normal(size=g.N * 6)
    b[:6] = 0.0
    x, _ = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
    xo, _ = O.linear_solve(og, d2, b, linear_solver=0)
    err = np.abs(x - xo).max() / np.abs(xo).max()
    print("%-24s N %6d E %7d  max rel err vs oracle %.3e %s" % (name, g.N, len(g.ia), err, "OK" if err < 1e-9 else "FAIL"), flush=True)
    return err < 1e-9


def timing(name, g, lm_iters=8):
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=lm_iters, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    t0 = time.time()
    prob.solver_begin(opt)
    t1 = time.time()
    tf = prob.time_kernel("front_factor", 5)
    ts = prob.time_kernel("front_solve", 5)
    t2 = time.time()
    ran, done = prob.solver_step(lm_iters)
    t3 = time.time()
    s = prob.solver_end()
    print("%-24s N %6d E %7d  begin %.1f ms  factor %.3f ms  solve %.3f ms  %d LM its %.1f ms  cost %.6e -> %.6e  kind %d levels %d maxfront %d flops %.3e -> %.2f TF/s" % (
        name, g.N, len(g.ia), 1e3 * (t1 - t0), tf, ts, ran, 1e3 * (t3 - t2), s.initial_cost, s.final_cost, s.c.factor_kind,
        s.factor_levels, s.c.factor_max_front, s.c.factor_flops, s.c.factor_flops / (tf * 1e-3) / 1e12), flush=True)
    return s


ok = True
ok &= check("manhattan 150", ds.manhattan_se3(150, 500, seed=2))
ok &= check("manhattan 400", ds.manhattan_se3(400, 1400, seed=7))
ok &= check("sphere 2x12x12", ds.sphere_layers(n_spheres=2, rings=12, per_ring=12))
ok &= check("manhattan 2000", ds.manhattan_se3(2000, 8000, seed=3))
if mode != "quick":
    ok &= check("manhattan 10k (C2)", ds.manhattan_se3())
    ok &= check("sphere 3x30x30", ds.sphere_layers(n_spheres=3, rings=30, per_ring=30))
timing("manhattan 2000", ds.manhattan_se3(2000, 8000, seed=3))
timing("manhattan 10k (C2)", ds.manhattan_se3())
if mode != "quick":
    timing("sphere x10 (C5)", ds.sphere_layers())
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
