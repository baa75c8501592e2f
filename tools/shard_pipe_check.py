"""Development aid: the sharded path on loopback virtual ranks, owner-only pipelined CG (default) against the replicated standard CG
(PGO_SHARD_PIPE=0) and against one rank.  usage (GPU box): python tools/shard_pipe_check.py [poses edges world its]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()


def solve_sharded(g, world, opt_kw):
    group = pkg.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = pkg.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            s = pkg.solve(pkg.SolverOptions(**opt_kw), prob)
            out[rank] = (s, poses)
        except Exception as e:
            errs.append(e)

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, errs
    pkg.loopback_destroy(group)
    return out


n, e, world, its = (int(a) for a in (sys.argv[1:5] + ["20000", "150000", "8", "6"][len(sys.argv) - 1:]))
g = ds.manhattan_se3(n, e, seed=20260930, loop_radius=3.0)
opt = dict(max_num_iterations=its, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=int(os.environ.get("CL", "2")))
# PGO_SHARD_TRACE_ONLY=1 (profiling): the sharded solve alone — no one-rank reference solve in the same kernel trace
trace_only = os.environ.get("PGO_SHARD_TRACE_ONLY", "0") == "1"
if not trace_only:
    prob, poses = pkg.problem_from_graph(g)
    ref = pkg.solve(pkg.SolverOptions(**opt), prob)
    print("one rank   cg", list(ref.iterations["linear_solver_iterations"]), "cost %.9e" % ref.final_cost)
for pipe in ("1", "0"):
    pipe = os.environ.get("PGO_SHARD_PIPE", pipe)      # (read once per process: only the first value counts -- run twice for both)
    os.environ["PGO_SHARD_PIPE"] = pipe
    out = solve_sharded(g, world, opt)
    s, p = out[0]
    print("pipe=%s w=%d cg" % (pipe, world), list(s.iterations["linear_solver_iterations"]), "cost %.9e" % s.final_cost,
          "" if trace_only else "max |dp| %.2e" % np.abs(p - poses).max(), "ranks identical", all(np.array_equal(out[0][1], o[1]) for o in out),
          "| cg_form %d, %.3f ms per LM iteration (solver clock, rank 0), exchange %s" % (
              s.cg_form, 1e3 * s.total_time_in_seconds / max(1, s.num_iterations - 1),
              "by the kernels" if os.environ.get("PGO_PEER_DIRECT", "0") == "1" else "host-enqueued all-gather (loopback copies)"))
    break
