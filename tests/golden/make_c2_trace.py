"""Generates tests/golden/c2_exact_trace.npz: the CPU oracle's LM trace (exact Cholesky steps, the reference's options
of finial.cpp:534-536: SPARSE_NORMAL_CHOLESKY, max_num_iterations 1000, Ceres 1.13 defaults otherwise) of BASELINE.json
configs[1] (Manhattan SE(3): 10 000 poses / 40 000 edges, seed 20260928) from the dead-reckoning start to the oracle's own
stop.  About two minutes of host time, too slow for the GPU suite, so it is a fixture; the GPU test
(tests/test_gpu_front.py::test_c2_exact_trace_to_convergence_matches_oracle_fixture) regenerates the same graph from the seed
and holds its trace to these numbers.  Run from the repo root: python tests/golden/make_c2_trace.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402
from oracle import oracle as O  # noqa: E402

ds = pgo_loader.datasets()
g = ds.manhattan_se3(10000, 40000, seed=20260928)
og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
poses, summ, trace = O.solve(og, O.default_options(max_num_iterations=1000, linear_solver=0), trace_capacity=1100)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c2_exact_trace.npz"), trace=trace,
                    initial_cost=summ.initial_cost, final_cost=summ.final_cost, n_poses=g.N, n_edges=g.E,
                    reason=summ.reason, termination_type=summ.termination_type,
                    num_successful_steps=summ.num_successful_steps, num_unsuccessful_steps=summ.num_unsuccessful_steps,
                    poses_head=poses[:64], poses_stride=poses[::50].copy(),
                    checksum_ia=int(np.asarray(g.ia, dtype=np.int64).sum()), checksum_meas=float(np.abs(g.meas).sum()))
print("iterations", len(trace), "cost", summ.initial_cost, "->", summ.final_cost, "reason", O.REASON.get(summ.reason),
      "seconds", summ.total_seconds)
