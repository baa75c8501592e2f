"""Development aid: PGO_VERBOSE phase timings of the exact-solver setup on C2 / C5 / C3 (second solve of each: warm process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    g = ds.manhattan_se3(10000, 40000)
elif which == "c5":
    g = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931)
else:
    kz = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
    offs = kz["cand_offsets"]
    cands = {int(key): kz["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(kz["cand_keys"])}
    g = ds.graph_from_candidates(kz["origin"], cands, seed=20260929)
for rep in range(2):
    if rep == 1:
        os.environ["PGO_VERBOSE"] = "1"
    prob, poses = pkg.problem_from_graph(g)
    t0 = time.perf_counter()
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=3, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY), prob)
    print("rep %d: solve %.2f ms (setup %.2f ms)" % (rep, 1e3 * (time.perf_counter() - t0), 1e3 * s.c.setup_time_in_seconds), flush=True)
