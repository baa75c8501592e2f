"""Development aid: the sharded symmetric-form CG on loopback virtual ranks of ONE GPU with the boundary exchange (default) against whole
segments per CG iteration (knob shard_boundary = 0): same answers bit for bit, wall time of the session, bytes per all-gather.
(Loopback copies on one device: the byte counts are what a real transport would carry, the times are not a scaling statement.)
usage (GPU box): python tools/shard_boundary_ab.py [world [LM iterations]]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
its = int(sys.argv[2]) if len(sys.argv) > 2 else 12
g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
opt = dict(max_num_iterations=its, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=0.1, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)


def ranks():
    group = pkg.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = pkg.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            prob.solver_begin(pkg.SolverOptions(**opt))
            prob.solver_step(2)
            t = time.perf_counter()
            ran, _ = prob.solver_step(its - 2)
            dt = time.perf_counter() - t
            out[rank] = (prob.solver_end(), poses, dt / max(1, ran))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(900)
    assert not errs, errs
    pkg.loopback_destroy(group)
    return out


cuts, rows_per = pkg.row_shard_cuts(g.N, g.ia, g.ib, world)
owner = np.searchsorted(np.asarray(cuts[1:]), np.arange(g.N), side="right")
cut = owner[g.ia] != owner[g.ib]
b = np.zeros(g.N, bool); b[g.ia[cut]] = True; b[g.ib[cut]] = True
bmax = max(int(b[cuts[r]:cuts[r + 1]].sum()) for r in range(world))
print("%d ranks: %.1f %% of the edges are cut, %.1f %% of the rows are boundary rows (most on a rank: %d of %d); per all-gather %d B of whole segments, %d B of boundary rows" % (
    world, 100 * cut.mean(), 100 * b.mean(), bmax, rows_per, world * (rows_per * 6 + 4) * 8, world * (((bmax + 1) & ~1) * 6 + 4) * 8))
res = {}
for mode in (1, 0, 1, 0):
    pkg.tuning_set("shard_boundary", mode)
    out = ranks()
    s = out[0][0]
    print("shard_boundary %d: cg_exchange %d, %.3f ms per LM iteration (rank 0, %d virtual ranks sharing one GPU), %d CG iterations, final cost %.9e" % (
        mode, s.cg_exchange, 1e3 * out[0][2], world, s.num_linear_solver_iterations, s.final_cost), flush=True)
    res.setdefault(mode, out)
pkg.tuning_set("shard_boundary", None)
a, c = res[1], res[0]
print("bit-identical between the two exchanges:", all(np.array_equal(x[1], y[1]) for x, y in zip(a, c)) and np.array_equal(a[0][0].iterations["cost"], c[0][0].iterations["cost"]))
