import os, sys
sys.path.insert(0, "/root/repo")
os.environ["PGO_BLOCK"] = "256"
import numpy as np
import pgo_loader
gpu = pgo_loader.load(); ds = pgo_loader.datasets()
g = ds.manhattan_se3(1500, 6000, seed=21)
opt = dict(max_num_iterations=40, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
def one():
    prob, poses = gpu.problem_from_graph(g)
    return gpu.solve(gpu.SolverOptions(**opt), prob)
def stepped(pattern, reset_after=None):
    prob, poses = gpu.problem_from_graph(g)
    prob.solver_begin(gpu.SolverOptions(**opt))
    if reset_after:
        prob.solver_step(reset_after); prob.solver_reset()
    done = False
    for n in pattern:
        if done: break
        ran, done = prob.solver_step(n)
    return prob.solver_end()
ref = one()
print("one-shot", len(ref.iterations), ref.final_cost, ref.cg_form)
for name, pat, rs in (("step(100)", (100,), None), ("reset+step(100)", (100,), 7), ("pauses", (1, 2, 1, 5, 3, 100), None), ("pauses1", (1,) * 60, None), ("reset+pauses", (1, 2, 1, 5, 3, 100), 7)):
    s = stepped(pat, rs)
    n = min(len(s.iterations), len(ref.iterations))
    same = [bool(np.array_equal(s.iterations[f][:n], ref.iterations[f][:n])) for f in ("cost", "linear_solver_iterations")]
    first = next((i for i in range(n) if s.iterations["cost"][i] != ref.iterations["cost"][i]), None)
    print(name, len(s.iterations), s.final_cost, same, "first diff at", first)
