"""The PCG's coarse level (options.pcg_coarse_aggregate) on the other BASELINE graphs: KITTI-00 replay (C1, a chain with 639 closures), KITTI-00 dense
candidates (C3), sphere x10 (C5): LM / CG iterations, final cost and wall time against the cluster Jacobi alone and the exact steps.
usage (GPU box): python tools/coarse_configs.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
c1 = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
offs = k["cand_offsets"]
cands = {int(key): k["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(k["cand_keys"])}
c3 = ds.graph_from_candidates(k["origin"], cands, seed=20260929)
c5 = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931)


def run(g, **kw):
    best = None
    for _ in range(2):
        prob, poses = gpu.problem_from_graph(g)
        t = time.perf_counter()
        s = gpu.solve(gpu.SolverOptions(**kw), prob)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]:
            best = (dt, s)
    return best


for name, g, aggs, its in (("C1 KITTI-00 replay", c1, (16, 32, 64), 1000), ("C3 KITTI-00 dense candidates", c3, (16, 32, 64), 1000), ("C5 sphere x10", c5, (64, 128), 60)):
    dt, s = run(g, max_num_iterations=its, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    print("%-30s exact steps          : cost %.6e, %3d LM iterations, %8.1f ms" % (name, s.final_cost, s.num_iterations - 1, 1e3 * dt), flush=True)
    ex = s.final_cost
    for agg in (0,) + tuple(aggs):
        try:
            dt, s = run(g, max_num_iterations=its, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=0.1, max_linear_solver_iterations=3000, pcg_coarse_aggregate=agg)
            print("%-30s %-21s: cost %.6e (%+.3f %%), %3d LM / %6d CG iterations, %8.1f ms" % (
                name, "cluster Jacobi" if agg == 0 else "+ coarse, agg %d" % agg, s.final_cost, 100 * (s.final_cost / ex - 1), s.num_iterations - 1, s.num_linear_solver_iterations, 1e3 * dt), flush=True)
        except Exception as e:  # noqa: BLE001
            print(name, agg, "ERR", str(e)[:150], flush=True)
