import os, sys
sys.path.insert(0, '/root/repo')
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000, seed=20260928)
for cl in (1, 2, 4):
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=2**30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cl, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    prob.solver_begin(opt)
    prob.solver_step(5)
    t = min(prob.time_kernel("uni_cg", 3) for _ in range(3))
    import time
    t0 = time.perf_counter(); ran, done = prob.solver_step(30); dt = time.perf_counter() - t0
    s = prob.solver_end()
    print("cluster", cl, "CG pair %.2f us" % (t * 1e3), "30 LM iterations %.3f ms each" % (dt * 1e3 / max(ran, 1)), "CG its", s.num_linear_solver_iterations, flush=True)
