"""Development aid: the symmetric-form CG product timed in isolation on BASELINE configs[3]'s graph (pgo_time_kernel 'sym_spmv',
5 x 100 launches).  For A/B comparisons of two builds on ONE box: copy both libpgo_hip.so variants into the tree and swap them
inside one gpurun command (boxes differ by 5 %).   usage (GPU box): python tools/spmv_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PGO_SYM", "1")
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
prob, _ = gpu.problem_from_graph(g)
prob.solver_begin(gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2))
prob.solver_step(2)
ts = sorted(prob.time_kernel("sym_spmv", 100) for _ in range(5))
print("sym_spmv min %.1f median %.1f us" % (ts[0] * 1e3, ts[2] * 1e3))
prob.solver_end()
