"""Development aid: BASELINE configs[3]'s graph (100 k poses / 1 M edges) solved on ONE GPU by the host-driven PCG path, with the CG
products read from the incidence-slot BSR (PGO_SYM=0) and from the symmetric tile form (PGO_SYM=1, the default above 600 k slots; knob sym_repack: its A/B):
ms per LM iteration, CG iterations, final cost.   usage (GPU box): python tools/c4_lm.py [steps [poses per preconditioner cluster]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

gpu = pgo_loader.load()
ds = pgo_loader.datasets()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cluster = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
for sym, rp in (("0", "0"), ("1", "1"), ("1", "0"), ("0", "0"), ("1", "1"), ("1", "0")):
    os.environ["PGO_SYM"] = sym
    gpu.tuning_set("sym_repack", int(rp))
    prob, poses = gpu.problem_from_graph(g)
    opt = gpu.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster, eta=0.1,
                            function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    prob.solver_begin(opt)
    prob.solver_step(2)
    t0 = time.perf_counter()
    ran, _ = prob.solver_step(steps)
    dt = time.perf_counter() - t0
    s = prob.solver_end()
    print("cluster %d PGO_SYM=%s sym_repack=%s: %.3f ms per LM iteration (%d iterations), %d CG iterations in the session, final cost %.9e" % (
        cluster, sym, rp, 1e3 * dt / max(1, ran), ran, s.num_linear_solver_iterations, s.final_cost), flush=True)
