"""Development aid: what tile size (rows per tile of the symmetric form) suits a SHARD-sized problem on a whole GPU — a one-rank
problem of (poses, edges) = one eighth of BASELINE configs[3], the one-launch CG iteration on the form (k_pipe_cg_sym) timed in situ
for the knob sym_rows = 32 .. 256.   usage (GPU box): python tools/sym_rows_probe.py [poses edges]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
n, e = (int(a) for a in (sys.argv[1:3] + ["12500", "125000"][len(sys.argv) - 1:]))
g = ds.manhattan_se3(n, e, seed=20260930, loop_radius=3.0)
os.environ["PGO_SYM"] = "1"
os.environ["PGO_NO_PIPELINE"] = "1"
for rows in (32, 48, 64, 96, 128, 256):
    gpu.tuning_set("sym_rows", rows)
    prob, _ = gpu.problem_from_graph(g)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2,
                                        function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    prob.solver_step(3)
    t = prob.time_kernel("sym_pipe_cg", 160)
    tl = prob.time_kernel("sym_linearize_lean", 50)
    s = prob.solver_end()
    print("rows %3d: k_pipe_cg_sym %.2f us per CG iteration, k_linearize_lean %.2f us (cg_form %d sym %d)" % (rows, 1e3 * t, 1e3 * tl, s.cg_form, s.sym_form), flush=True)
