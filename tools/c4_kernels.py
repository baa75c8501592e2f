"""Development aid: the hot kernels timed in isolation (pgo_time_kernel, HIP events) on BASELINE configs[3]'s graph (100 k poses /
1 M edges, one GPU) and on configs[1]'s (10 k / 40 k), with the fraction of the 8 TB/s HBM peak on SURVEY 8d's algorithmic bytes.
usage (GPU box): [PGO_LIN_VARIANT=n] python tools/c4_kernels.py [c4|c2|both]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
cases = []
if which in ("c4", "both"):
    cases.append(("c4", ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)))
if which in ("c2", "both"):
    cases.append(("c2", ds.manhattan_se3(10000, 40000, seed=20260928)))
for name, g in cases:
    N, E = g.N, len(g.ia)
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    prob.solver_begin(opt)
    prob.solver_step(2)
    out = []
    for k, nbytes in (("linearize", 640 * E + 392 * N), ("pcg_spmv", (N + E) * 288 + 2 * N * 48), ("sym_spmv", (N + E) * 288 + 2 * N * 48),
                      ("sym_linearize_rows", 640 * E + 392 * N), ("sym_linearize_lean", 640 * E + 392 * N), ("evaluate", 976 * E + 56 * N)):
        reps = 50 if name == "c4" else 300
        t = min(prob.time_kernel(k, reps) for _ in range(3))
        out.append("%s %.1f us %.0f GB/s frac %.3f" % (k, t * 1e3, nbytes / (t * 1e-3) / 1e9, nbytes / (t * 1e-3) / 1e9 / 8000.0))
    prob.solver_end()
    print(name, "variant", os.environ.get("PGO_LIN_VARIANT", "0"), "|", " | ".join(out), flush=True)
