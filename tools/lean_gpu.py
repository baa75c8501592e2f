"""Development aid: k_linearize_lean against k_linearize_symout — largest relative difference of everything they write
(pgo_time_kernel 'sym_lean_check') on a few graphs, and the launch times on BASELINE configs[3]'s graph.
usage (GPU box): [PGO_LEAN_WAVES=2|3|4] python tools/lean_gpu.py [check|time|both]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PGO_SYM", "1")
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def session(g, **kw):
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, **kw)
    prob.solver_begin(opt)
    prob.solver_step(2)
    return prob, poses


if which in ("check", "both"):
    gb = ds.manhattan_se3(2500, 9000, seed=8)
    rng = np.random.default_rng(4)
    L = np.zeros((len(gb.ia), 6, 6))          # block-diagonal information: W_pp, W_rr full, no coupling
    for blk in (slice(0, 3), slice(3, 6)):
        A = rng.normal(size=(len(gb.ia), 3, 3)) * 0.3 + 3.0 * np.eye(3)
        L[:, blk, blk] = A
    gb.sqrt_info = L
    graphs = {"manhattan (diagonal information)": ds.manhattan_se3(4000, 16000, seed=21),
              "sphere": ds.sphere_layers(n_spheres=2, rings=20, per_ring=20),
              "fat rows": ds.manhattan_se3(600, 3000, seed=5, loop_radius=6.0),
              "identity information": ds.manhattan_se3(1500, 5000, seed=3, identity_information=True),
              "block-diagonal information": gb}
    for name, g in graphs.items():
        prob, _ = session(g)
        print("%-36s N %6d E %7d   largest relative difference %.3e" % (name, g.N, len(g.ia), prob.time_kernel("sym_lean_check", 1)), flush=True)
        prob.solver_end()
if which in ("time", "both"):
    g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
    N, E = g.N, len(g.ia)
    prob, _ = session(g)
    print("c4 check: largest relative difference %.3e" % prob.time_kernel("sym_lean_check", 1), flush=True)
    nbytes = 640 * E + 392 * N
    for k in ("sym_linearize_rows", "sym_linearize_lean", "linearize"):
        ts = sorted(prob.time_kernel(k, 50) for _ in range(5))
        print("%-22s min %.1f median %.1f us   frac %.3f (median)" % (k, ts[0] * 1e3, ts[2] * 1e3, nbytes / (ts[2] * 1e-3) / 1e9 / 8000.0), flush=True)
    prob.solver_end()
