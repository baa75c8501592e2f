"""Randomised sweep of the symmetric tile form (csrc/pgo_sym.*) against the incidence-slot kernels on the same GPU: random graphs
(lattice walks, random chords with duplicate edges, hubs with hundreds of incidences, tiny graphs), random information kinds, constant
blocks, losses, tile caps — whole LM solves (PCG, host-driven) and single linear systems.  Prints every mismatch.
usage (GPU box): python tools/fuzz_sym.py [n_cases] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
os.environ["PGO_NO_PIPELINE"] = "1"       # the host-driven CG is the one that reads the symmetric form


def random_case(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(2, 1500 if kind else 3000))
    if kind == 0:
        possible = max(0, n - 21) * max(0, n - 20) // 2
        e = n - 1 + int(rng.integers(0, min(4 * n + 2, possible) + 1))
        g = ds.manhattan_se3(n, e, seed=int(rng.integers(1 << 30)), identity_information=bool(rng.integers(0, 2)))
    else:
        poses = np.zeros((n, 7))
        poses[:, :3] = rng.normal(0, 2.0, (n, 3))
        q = rng.normal(size=(n, 4))
        poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
        extra = int(rng.integers(0, 4 * n))
        xa, xb = rng.integers(0, n, extra), rng.integers(0, n, extra)
        if kind == 2 and n > 10:          # hubs: a few poses collect hundreds of incidences (rows spanning several chunks)
            hubs = rng.integers(0, n, 3)
            xb[: extra // 2] = rng.choice(hubs, extra // 2)
        if kind == 3 and extra > 4:       # duplicate edges
            xa[: extra // 3] = xa[0]; xb[: extra // 3] = xb[0]
        ia = np.concatenate([np.arange(1, n), xa]).astype(np.int32)
        ib = np.concatenate([np.arange(0, n - 1), xb]).astype(np.int32)
        keep = ia != ib
        ia, ib = ia[keep], ib[keep]
        meas = ds.relative_pose(poses[ia], poses[ib])
        meas[:, :3] += rng.normal(0, 0.05, (len(ia), 3))
        meas[:, 3:] = ds.qmul(ds.qexp_half(rng.normal(0, 0.01, (len(ia), 3))), meas[:, 3:])
        info = None
        t = rng.integers(0, 3)
        if t == 1:
            info = np.repeat(np.diag(rng.uniform(0.5, 3.0, 6)).reshape(1, 36), len(ia), axis=0)
        elif t == 2:
            A = rng.normal(size=(len(ia), 6, 6))
            info = (np.linalg.cholesky(A @ np.transpose(A, (0, 2, 1)) + 6 * np.eye(6)) * 0.3).reshape(-1, 36)
        start = poses.copy()
        start[1:, :3] += rng.normal(0, 0.3, (n - 1, 3))
        start[1:, 3:] = ds.qmul(ds.qexp_half(rng.normal(0, 0.05, (n - 1, 3))), start[1:, 3:])
        g = ds.PoseGraphData(start, ia, ib, meas, info)
    cmask = np.zeros(g.N, dtype=np.uint8)
    cmask[0] = 3
    for v in rng.integers(0, g.N, int(rng.integers(0, 3))):
        cmask[v] = rng.integers(1, 4)
    return (g, cmask, int(rng.integers(0, 5)), float(rng.uniform(0.3, 3.0)), int(rng.choice([1, 2, 4])), int(rng.choice([8, 16, 33, 64, 256])), bool(rng.integers(0, 2)),
            int(rng.choice([1, 0])))       # pcg_form: 1 = Ceres' refreshed CG on the form (k_spmv_sym<0>), 0 = the library's choice (r06: the one-launch pipelined iteration, k_pipe_cg_sym, where the form is the session's storage)


def solve(g, cmask, loss, loss_a, cluster, sym, rows, repack, form):
    os.environ["PGO_SYM"] = "1" if sym else "0"
    pkg.tuning_set("sym_rows", rows)
    pkg.tuning_set("sym_repack", 1 if repack else None)
    prob, poses = pkg.problem_from_graph(g, loss=loss, loss_a=loss_a, constant_first=False)
    for v in np.nonzero(cmask)[0]:
        prob.set_pose_constant(int(v), int(cmask[v]))
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=8, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=cluster, pcg_form=form), prob)
    return s, poses


def main(n_cases=100, first=0):
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n_cases):
        g, cmask, loss, loss_a, cluster, rows, repack, form = random_case(seed)
        try:
            a, pa = solve(g, cmask, loss, loss_a, cluster, False, rows, False, form)
            b, pb = solve(g, cmask, loss, loss_a, cluster, True, rows, repack, form)
            c, pc = solve(g, cmask, loss, loss_a, cluster, True, rows, repack, form)
        except Exception as exc:   # noqa: BLE001
            print("seed", seed, "EXCEPTION", exc, flush=True)
            bad += 1
            continue
        n = min(len(a.iterations), len(b.iterations), 4)
        why = []
        if not (np.array_equal(pb, pc) and np.array_equal(b.iterations["cost"], c.iterations["cost"])):
            why.append("not reproducible")
        if list(a.iterations["step_is_successful"][:n]) != list(b.iterations["step_is_successful"][:n]):
            why.append("decisions differ in the first %d iterations" % n)
        if not np.allclose(a.iterations["cost"][:n], b.iterations["cost"][:n], rtol=1e-7, atol=1e-12):
            why.append("costs differ: %s vs %s" % (a.iterations["cost"][:n], b.iterations["cost"][:n]))
        fixed = np.nonzero(cmask == 3)[0]
        if not (np.isfinite(pb).all() and np.array_equal(pb[fixed], g.poses[fixed])):
            why.append("constant pose moved / non-finite")
        if why:
            bad += 1
            print("seed", seed, "N", g.N, "E", g.E, "rows", rows, "repack", repack, "cluster", cluster, "loss", loss, "form", form, "cg_form", b.cg_form, ":", "; ".join(why), flush=True)
    print("%d cases, %d bad, %.1f s" % (n_cases, bad, time.time() - t0), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
