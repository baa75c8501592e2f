"""Randomised sweep of the resident universal stream (csrc/pgo_uni_resident.h, pcg_form 3) against the fused one (pcg_form 2) on the same GPU:
random Manhattan graphs from a handful of poses up to grids that fill the chip (and beyond: those must take the fused stream), work-group
sizes 64 / 128 / 256 / the library's choice, 6x6 and 12x12 Jacobi blocks, identity / diagonal information, with and without Huber.
Same decisions and CG counts, costs to 1e-7 (rejected candidates 1e-5), or the case is printed.
usage (GPU box): python tools/fuzz_resident.py [n_cases] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()


def run(g, form, cl, its, eta):
    prob, poses = pkg.problem_from_graph(g)
    s = pkg.solve(pkg.SolverOptions(max_num_iterations=its, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=cl, pcg_form=form, eta=eta), prob)
    return s, poses


def main(n_cases, seed0):
    bad = resident = 0
    t0 = time.time()
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        n = int(rng.choice([rng.integers(3, 60), rng.integers(60, 2000), rng.integers(2000, 9000), rng.integers(9000, 16000)]))
        possible = max(0, n - 21) * max(0, n - 20) // 2
        e = n - 1 + int(rng.integers(0, min(4 * n + 2, possible) + 1))
        block = [None, 64, 128, 256][int(rng.integers(0, 4))]
        cl = int(rng.integers(1, 3))
        ident = bool(rng.integers(0, 2))
        eta = float(rng.choice([0.1, 0.3, 0.02]))
        its = int(rng.integers(3, 12))
        if block is None: os.environ.pop("PGO_BLOCK", None)
        else: os.environ["PGO_BLOCK"] = str(block)
        g = ds.manhattan_se3(n, e, seed=int(rng.integers(1 << 30)), identity_information=ident)
        a, pa = run(g, 2, cl, its, eta)
        b, pb = run(g, 3, cl, its, eta)
        resident += b.cg_form == 4
        ok = (list(a.iterations["step_is_successful"]) == list(b.iterations["step_is_successful"]) and
              list(a.iterations["linear_solver_iterations"]) == list(b.iterations["linear_solver_iterations"]) and
              np.allclose(a.iterations["cost"], b.iterations["cost"], rtol=np.where(a.iterations["step_is_successful"] > 0, 1e-7, 1e-5)) and   # (a rejected candidate behind a CG run of hundreds of iterations: 1.6e-6 seen)
              np.abs(pa - pb).max() < 1e-5 and a.termination_type == b.termination_type)
        if not ok:
            n_it = min(len(a.iterations), len(b.iterations))
            rel = np.abs(a.iterations["cost"][:n_it] - b.iterations["cost"][:n_it]) / np.abs(a.iterations["cost"][:n_it])
            print("    decisions %s, CG counts %s, worst cost rel %.2e at iteration %d (successful %s), termination %d / %d" % (
                list(a.iterations["step_is_successful"]) == list(b.iterations["step_is_successful"]),
                list(a.iterations["linear_solver_iterations"]) == list(b.iterations["linear_solver_iterations"]),
                rel.max(), int(rel.argmax()), bool(a.iterations["step_is_successful"][int(rel.argmax())]), a.termination_type, b.termination_type))
        if not ok or case % 10 == 0:
            print("%s case %d (seed %d): %d poses %d edges block %s cluster %d identity %d eta %g its %d: cg_form %d / %d, CG %d / %d, cost %.10e / %.10e, |dp| %.1e" % (
                "ok " if ok else "BAD", case, seed0 + case, n, e, block, cl, ident, eta, its, a.cg_form, b.cg_form, a.num_linear_solver_iterations,
                b.num_linear_solver_iterations, a.final_cost, b.final_cost, np.abs(pa - pb).max()), flush=True)
        bad += not ok
    print("%d cases, %d ran the resident stream, %d mismatches, %.0f s" % (n_cases, resident, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1000))
