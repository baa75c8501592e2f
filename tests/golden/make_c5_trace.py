"""Generates tests/golden/c5_exact_trace.npz: the CPU oracle's LM trace (exact Cholesky steps, reference options) of
BASELINE.json configs[4] (sphere x10: 25 000 poses / 250 000 edges, seed 20260931) over 6 iterations.  The oracle needs
about 20 s per iteration at this size, which is too slow for the GPU test suite; the GPU test
(tests/test_gpu_front.py::test_c5_lm_trace_matches_oracle_fixture) regenerates the same graph from the seed and compares its
trace with these numbers.  Run from the repo root: python tests/golden/make_c5_trace.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402
from oracle import oracle as O  # noqa: E402

ds = pgo_loader.datasets()
g = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931)
og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
poses, summ, trace = O.solve(og, O.default_options(max_num_iterations=6, linear_solver=0))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_exact_trace.npz"), trace=trace,
                    initial_cost=summ.initial_cost, final_cost=summ.final_cost, n_poses=g.N, n_edges=g.E,
                    poses_head=poses[:64], checksum_ia=int(np.asarray(g.ia, dtype=np.int64).sum()),
                    checksum_meas=float(np.abs(g.meas).sum()))
print("iterations", len(trace), "cost", summ.initial_cost, "->", summ.final_cost)
