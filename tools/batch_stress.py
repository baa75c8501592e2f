"""Race hunt: many copies of one graph as ONE batched solve put thousands of workgroups of the factorisation kernels in flight at
once (several per CU, waves of a workgroup drifting apart).  Identical components must come out equal iteration by iteration —
to the rounding of the linearisation's lane-pair sums (where a row's pairs fall depends on the parity of its first slot inside the
union: a handful of distinct traces 1e-13 apart; a race moves the 7th digit) — and every repeat must reproduce the first bit for bit.
Run under PGO_FRONT=1 / PGO_SFRONT=1 / defaults to cover the three exact solvers.
usage: python tools/batch_stress.py [copies] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
k = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 48
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
graphs = {"kitti00": ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None),
          "manhattan_600_1500": ds.manhattan_se3(600, 1500, seed=7),
          "sphere_2x12x12": ds.sphere_layers(n_spheres=2, rings=12, per_ring=12),
          "manhattan_2000_5000": ds.manhattan_se3(2000, 5000, seed=9)}
opt = pkg.SolverOptions(max_num_iterations=25, linear_solver_type=pkg.SPARSE_NORMAL_CHOLESKY)
bad = 0
for name, g in graphs.items():
    first = None
    for rep in range(repeats):
        pairs = [pkg.problem_from_graph(g) for _ in range(copies)]
        sums = pkg.solve_batch(opt, [p for p, _ in pairs])
        traces = [np.array(s.iterations["cost"], dtype=np.float64) for s in sums]
        same_len = len({len(t) for t in traces}) == 1
        spread = max(float(np.max(np.abs(t - traces[0]) / np.abs(traces[0]))) for t in traces) if same_len else float("inf")
        pspread = max(float(np.abs(p - pairs[0][1]).max()) for _, p in pairs)
        key = (tuple(t.tobytes() for t in traces), tuple(p.tobytes() for _, p in pairs))
        if first is None:
            first = key
        ok = same_len and spread <= 1e-11 and pspread <= 1e-9 * max(1.0, float(np.abs(pairs[0][1]).max())) and key == first
        bad += not ok
        print("%-22s rep %d: kind %d, %2d iterations, %d distinct cost traces (relative spread %.1e), poses within %.1e, %s %s" % (
            name, rep, sums[0].c.factor_kind, len(sums[0].iterations) - 1, len({t.tobytes() for t in traces}), spread, pspread,
            "same bits as rep 0" if key == first else "DIFFERS from rep 0", "" if ok else "<-- MISMATCH"), flush=True)
print("mismatching batches:", bad)
