"""What rocprofv3 wraps for the C4-size evidence (profiles/rNN_c4_*): BASELINE configs[3]'s graph (100 k poses / 1 M edges) on one
GPU, `steps` LM iterations of the bench's PCG policy through the host-driven loop (graphs this large keep it), then the isolated
kernels (pgo_time_kernel: 30 launches each of linearize / pcg_spmv / evaluate) so that every kernel of SURVEY 8d appears in the trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
prob, poses = gpu.problem_from_graph(g)
opt = gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2,
                        function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
prob.solver_begin(opt)
ran, done = prob.solver_step(steps)
# (a PCG session of this size keeps the normal equations in the symmetric tile form: its LM loop runs the *_sym kernels; the
# incidence-slot kernels are timed next to them for the comparison)
for k in ("sym_pipe_cg", "sym_spmv", "sym_linearize_lean", "sym_linearize_rows", "linearize", "pcg_spmv", "evaluate"):
    prob.time_kernel(k, 30)
s = prob.solver_end()
print("C4 %d poses / %d edges: %d LM iterations, %d CG iterations, cost %.6e -> %.6e" % (
    g.N, len(g.ia), ran, s.num_linear_solver_iterations, s.initial_cost, s.final_cost))
