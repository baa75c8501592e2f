"""Factor + solve time of the two GPU factorisations (enumerated 6x6 pairs vs multifrontal) per configuration.
usage: python tools/front_vs_direct.py [c1 c3 c2 c5 m2000 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()


def graph(name):
    if name == "c2": return ds.manhattan_se3()
    if name == "c5": return ds.sphere_layers()
    if name == "m2000": return ds.manhattan_se3(2000, 8000, seed=3)
    if name == "s3": return ds.sphere_layers(n_spheres=3, rings=30, per_ring=30)
    k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
    if name == "c1": return ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    offs = k["cand_offsets"]
    cands = {int(key): k["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(k["cand_keys"])}
    return ds.graph_from_candidates(k["origin"], cands, seed=20260929)


for name in (sys.argv[1:] or ["c1", "c3", "m2000", "c2"]):
    g = graph(name)
    for mode in ("0", "1"):
        os.environ["PGO_FRONT"] = mode
        os.environ["PGO_DIRECT_MAX_STEPS"] = "1e9"
        os.environ["PGO_DIRECT_HYBRID_STEPS"] = "1e9"
        os.environ["PGO_DIRECT_MAX_PAIRS"] = "100000000"
        prob, poses = gpu.problem_from_graph(g)
        t0 = time.time()
        try:
            prob.solver_begin(gpu.SolverOptions(max_num_iterations=3, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
            t1 = time.time()
            ms = prob.time_kernel("direct", 10)
            s = prob.solver_end()
            print("%-6s N %6d E %7d  %-12s begin %7.1f ms  factor+solve %8.3f ms  (kind %d, levels %d, blocks %d)" % (
                name, g.N, len(g.ia), "multifrontal" if mode == "1" else "pair lists", 1e3 * (t1 - t0), ms, s.c.factor_kind, s.factor_levels, s.factor_nnz_blocks), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-6s %s: %s" % (name, mode, e), flush=True)
