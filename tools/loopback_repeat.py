import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
world = int(sys.argv[1]); reps = int(sys.argv[2]); nit = int(sys.argv[3])
g = ds.manhattan_se3(1001, 3700, seed=31)
prob0, poses0 = pkg.problem_from_graph(g)
ref = pkg.solve(pkg.SolverOptions(max_num_iterations=nit, linear_solver_type=pkg.BLOCK_JACOBI_PCG), prob0)
refk = tuple(int(x) for x in ref.iterations["linear_solver_iterations"])
bad = 0
for rep in range(reps):
    group = pkg.loopback_create(world)
    out = [None] * world
    def run(rank):
        try:
            prob, poses = pkg.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            s = pkg.solve(pkg.SolverOptions(max_num_iterations=nit, linear_solver_type=pkg.BLOCK_JACOBI_PCG), prob)
            out[rank] = (tuple(int(x) for x in s.iterations["linear_solver_iterations"]), repr(s.final_cost))
        except Exception as e:
            out[rank] = ("ERR", str(e)[:60])
    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in ts]; [t.join(60) for t in ts]
    ok = all(o is not None and o[0] == refk for o in out) and len(set(out)) == 1
    bad += (not ok)
    if not ok: print("rep", rep, "BAD", out)
print("world", world, "reps", reps, "bad", bad, "env sync", os.environ.get("PGO_LOOPBACK_SYNC"))
