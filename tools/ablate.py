import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000)
prob, poses = pkg.problem_from_graph(g)
opt = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.BLOCK_JACOBI_PCG, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
prob.solver_begin(opt); prob.solver_step(3)
for dbg in [0, 1, 2, 3, 4, 7]:
    pkg.tuning_set("debug", dbg)
    print("debug", dbg, "pcg_spmv us", round(prob.time_kernel("pcg_spmv", 500) * 1e3, 2), "spmv(plain) us", round(prob.time_kernel("spmv", 500) * 1e3, 2))
pkg.tuning_set("debug", None)
for k in ["empty", "touch", "pcg_update", "cost", "linearize"]:
    print(k, round(prob.time_kernel(k, 500) * 1e3, 2))
print("pcg_graph per-iteration us", round(prob.time_kernel("pcg_graph", 5) * 1e3, 2))
