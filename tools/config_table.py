"""Per-configuration results (BASELINE.json configs C1..C5) on ONE GPU: wall time of a solve with the stated options,
iterations, final cost, time per LM iteration.  The oracle is NOT run here (parity for each configuration is what
tests/ checks); this prints the timing table of DESIGN.md section 7.  usage: python tools/config_table.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
G = os.path.join(ROOT, "tests", "golden")


def run(name, g, opt, repeat=2):
    best = None
    for _ in range(repeat):
        prob, poses = pkg.problem_from_graph(g)
        t = time.perf_counter()
        s = pkg.solve(opt, prob)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]:
            best = (dt, s)
    dt, s = best
    its = max(1, s.num_iterations)
    print("| %s | %d / %d | %s | %d | %d | %.6e -> %.6e | %.1f ms (%.1f ms setup) | %.3f ms |" % (
        name, g.N, g.E, {0: "direct", 1: "PCG", 2: "PCG to 1e-13", 3: "direct or PCG to 1e-13 per iteration"}[s.linear_solver_used], s.num_iterations,
        s.num_linear_solver_iterations, s.initial_cost, s.final_cost, 1e3 * dt, 1e3 * s.setup_time_in_seconds,
        1e3 * (s.total_time_in_seconds) / its), flush=True)


def main():
    print("| config | poses / edges | linear solver | LM its | CG its | cost | wall (pgo_solve, host buffers in/out) | per LM iteration |")
    print("|---|---|---|---|---|---|---|---|")
    k = np.load(os.path.join(G, "kitti00.npz"))
    c1 = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    exact = pkg.SolverOptions(max_num_iterations=1000, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
    pcg = lambda n: pkg.SolverOptions(max_num_iterations=n, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    run("C1 KITTI-00 replay, reference options (exact)", c1, exact)
    run("C1 KITTI-00 replay, PCG", c1, pcg(1000))
    c2 = ds.manhattan_se3(10000, 40000)
    run("C2 Manhattan, PCG, 25 iterations", c2, pcg(25))
    run("C2 Manhattan, PCG, to convergence (<=1000)", c2, pcg(1000), repeat=1)
    run("C2 Manhattan, exact request", c2, exact, repeat=1)
    offs = k["cand_offsets"]
    cands = {int(key): k["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(k["cand_keys"])}
    c3 = ds.graph_from_candidates(k["origin"], cands, seed=20260929)
    run("C3 KITTI-00 dense candidates, exact", c3, exact)
    run("C3 KITTI-00 dense candidates, PCG", c3, pcg(1000))
    c4 = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
    run("C4 synthetic large (one GPU), PCG, 25 iterations", c4, pcg(25), repeat=1)
    c5 = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931)
    run("C5 sphere x10, PCG, 30 iterations", c5, pcg(30), repeat=1)
    run("C5 sphere x10, exact request, 30 iterations", c5, pkg.SolverOptions(max_num_iterations=30, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY), repeat=1)


if __name__ == "__main__":
    main()
