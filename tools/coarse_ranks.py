"""Coarse level of the PCG on virtual ranks of one GPU (loopback transport): CG work and final cost per world size against the one-rank
solve and the exact path.  `python tools/coarse_ranks.py [poses] [edges] [agg] [max LM iterations] [worlds, comma separated]`"""
import os
import sys
import threading

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
gpu.build()
ds = pgo_loader.datasets()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4800
agg = int(sys.argv[3]) if len(sys.argv) > 3 else 32
max_it = int(sys.argv[4]) if len(sys.argv) > 4 else 400
worlds = [int(w) for w in sys.argv[5].split(",")] if len(sys.argv) > 5 else [2, 3, 4, 8]
g = ds.manhattan_se3(N, E, seed=20260930, loop_radius=3.0) if N >= 100000 else ds.manhattan_se3(N, E, seed=5)      # (100 000 / 1 000 000: BASELINE configs[3])
opt = dict(max_num_iterations=max_it, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_coarse_aggregate=agg, eta=0.1)


def ranks(world):
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = gpu.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            out[rank] = (gpu.solve(gpu.SolverOptions(**opt), prob), poses)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    gpu.loopback_destroy(group)
    return out


if N <= 20000:
    prob, _ = gpu.problem_from_graph(g)
    exact = gpu.solve(gpu.SolverOptions(max_num_iterations=400, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), prob)
    print(f"exact: cost {exact.final_cost:.6f} in {exact.num_iterations} LM iterations")
else:       # (too long for a probe: the tight-eta cluster Jacobi stands in)
    prob, _ = gpu.problem_from_graph(g)
    exact = gpu.solve(gpu.SolverOptions(max_num_iterations=max_it, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=0.1), prob)
    print(f"cluster Jacobi alone, eta 0.1: cost {exact.final_cost:.6f} in {exact.num_iterations} LM / {exact.num_linear_solver_iterations} CG iterations, {exact.total_time_in_seconds * 1e3:.1f} ms")
prob, _ = gpu.problem_from_graph(g)
one = gpu.solve(gpu.SolverOptions(**opt), prob)
print(f"1 rank: cost {one.final_cost:.6f} ({one.final_cost / exact.final_cost - 1:+.2e}), {one.num_iterations} LM, {one.num_linear_solver_iterations} CG, coarse {one.coarse_level}, {one.total_time_in_seconds * 1e3:.1f} ms")
for world in worlds:
    out = ranks(world)
    s = out[0][0]
    same = all((x == out[0][1]).all() for _, x in out)
    print(f"{world} ranks: cost {s.final_cost:.6f} ({s.final_cost / exact.final_cost - 1:+.2e}), {s.num_iterations} LM, {s.num_linear_solver_iterations} CG, coarse {s.coarse_level}, "
          f"exchange {s.cg_exchange}, identical {same}, {s.total_time_in_seconds * 1e3:.1f} ms, termination {s.termination_type}")
    print("   CG per LM:", list(s.iterations["linear_solver_iterations"][:25]))
