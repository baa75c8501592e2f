"""Scratch GPU check used during development: smoke + kernel timings on the bench workload."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
from oracle import oracle as O
import __graft_entry__ as ge
t = time.time(); ge.smoke(); print("smoke", time.time() - t)
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 40000)
t = time.time(); g = ds.manhattan_se3(n, e); print("gen", time.time() - t)
prob, poses = pkg.problem_from_graph(g)
opt = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.BLOCK_JACOBI_PCG, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
t = time.time(); prob.solver_begin(opt); print("begin", time.time() - t)
t = time.time(); ran, done = prob.solver_step(5); print("5 steps", time.time() - t)
t = time.time(); ran, done = prob.solver_step(20); dt = time.time() - t; print("20 steps", dt, "per step ms", dt / 20 * 1e3)
for k in ["linearize", "spmv", "pcg_spmv", "pcg_update", "pcg_iteration", "cost", "evaluate"]:
    print(k, "avg ms", prob.time_kernel(k, 200))
s = prob.solver_end()
print(s.full_report())
