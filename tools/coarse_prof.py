"""Profiling driver: BASELINE configs[1] with the coarse level, a fixed number of LM iterations (for rocprofv3 --kernel-trace).
usage: python tools/coarse_prof.py [aggregate [LM iterations]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
agg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
its = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = ds.manhattan_se3(10000, 40000, seed=20260928)
prob, poses = gpu.problem_from_graph(g)
prob.solver_begin(gpu.SolverOptions(max_num_iterations=10 ** 6, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg, eta=0.1,
                                    function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
prob.solver_step(2)
t = time.perf_counter(); ran, _ = prob.solver_step(its); dt = time.perf_counter() - t
s = prob.solver_end()
cg = sum(s.iterations["linear_solver_iterations"][3:3 + ran])
print("aggregates of %d: %.3f ms per LM iteration over %d, %.1f CG iterations each, %.2f us per CG iteration all in" % (agg, 1e3 * dt / ran, ran, cg / ran, 1e6 * dt / max(1, cg)))
