import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import fuzz_parity as F
from oracle import oracle as O
pkg = F.pkg
for seed in (91011, 91067):
    g, cmask, loss, loss_a, exact, cluster = F.random_case(seed)
    res = {}
    for name, env in (("host", {"PGO_NO_PIPELINE": "1"}), ("seq", {"PGO_UNI": "0"}), ("uni", {})):
        for k in ("PGO_NO_PIPELINE", "PGO_UNI"): os.environ.pop(k, None)
        pkg.tuning_set("pipeline_pcg", 1 if name == "seq" else None)
        os.environ.update(env)
        prob, poses = pkg.problem_from_graph(g, loss=loss, loss_a=loss_a, constant_first=False)
        for v in np.nonzero(cmask)[0]: prob.set_pose_constant(int(v), int(cmask[v]))
        s = pkg.solve(pkg.SolverOptions(max_num_iterations=12, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster), prob)
        res[name] = s
        print(seed, name, len(s.iterations), s.termination_type, s.message, list(s.iterations["step_is_successful"]), "%.15e" % s.final_cost)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info, cmask)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=12, linear_solver=1, pcg_cluster=cluster, loss_kind=loss, loss_a=loss_a))
    print(seed, "oracle", len(otr), osum.termination_type, [int(x) for x in otr[:, 8]], "%.15e" % osum.final_cost)
    print("   cost rel diff seq vs oracle", np.abs(res["seq"].iterations["cost"] - otr[:, 1]).max() / otr[0, 1])
