"""The reference's own configuration (C1: KITTI-00 replay, SPARSE_NORMAL_CHOLESKY -> the GPU block Cholesky) solved a few times,
for `rocprofv3 --kernel-trace` (summary kept as profiles/r01_exact_c1_kernel_stats.csv).  The oracle is not run.
usage: rocprofv3 --kernel-trace -d gpurun_out/prof_exact -o exact -- python tools/profile_exact.py [c1|c2|c3|sphere] [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
if which == "c3":
    offs = k["cand_offsets"]
    cands = {int(key): k["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(k["cand_keys"])}
    g = ds.graph_from_candidates(k["origin"], cands, seed=20260929)
elif which == "sphere":
    g = ds.sphere_layers(n_spheres=1, rings=50, per_ring=50, n_edges=25000, seed=20260931)
elif which == "c2":
    g = ds.manhattan_se3(10000, 40000)
else:
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
opt = pkg.SolverOptions(max_num_iterations=1000 if which in ("c1", "c3") else 30, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    prob, poses = pkg.problem_from_graph(g)
    t = time.perf_counter()
    s = pkg.solve(opt, prob)
    print("%s run %d: %d poses %d edges, solver %d, %d LM iterations, cost %.9e -> %.9e, %.2f ms (setup %.2f ms)" % (
        which, r, g.N, g.E, s.linear_solver_used, s.num_iterations, s.initial_cost, s.final_cost,
        1e3 * (time.perf_counter() - t), 1e3 * s.setup_time_in_seconds), flush=True)
