"""The sharded path's collective, measured where it can be on a one-GPU box: pgo_time_kernel('exchange') = the RCCL all-gather of
the q segment enqueued back to back on the solver stream (HIP events), world size 1 (ncclCommInitRank with one rank), for the
segment sizes of BASELINE configs[1] and configs[3] over 8 ranks.  This is the enqueue + launch floor of an ncclAllGather on this
stack — the xGMI transfer itself needs the driver's 8-GPU node.
usage (GPU box): python tools/exchange_latency.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
out = {}
for name, n, e in (("c2_10k_40k", 10000, 40000), ("c4_over_8_ranks_12500_rows", 12500, 125000)):
    g = ds.manhattan_se3(n, e, seed=7)
    prob, poses = gpu.problem_from_graph(g)
    prob.comm_init(gpu.comm_unique_id(), 0, 1)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=100, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2))
    prob.solver_step(1)
    ms = min(prob.time_kernel("exchange", 200) for _ in range(3))
    prob.solver_end()
    out[name] = {"segment_doubles": 6 * n, "us_per_all_gather_world_1": round(1e3 * ms, 2)}
print(json.dumps(out))
