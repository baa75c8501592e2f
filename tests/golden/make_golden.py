"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only (it reads
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Fixtures are DATA: inputs + expected outputs.
  * kitti00.npz        — data files the reference holds (REF = /root/reference/src/POSE_GRAPH_CERES_PLUS):
                         result/trajectory/trajectory_origin.txt (solver input poses),
                         result/trajectory/trajectory_update_y_not_constant.txt (Ceres 1.13 output, information only),
                         result/Edges/edges_for_loop.txt (accepted loop-edge id pairs),
                         config/Edge_Candidates_index.txt (as flattened ids + offsets + sha256 of the file text);
                         plus the replay graph built from them (odometry recomputed in float32 as
                         finial.cpp:214-215 does; loop measurements are documented STAND-INS taken from the
                         committed output trajectory because the reference never recorded its PnP results).
  * kitti00_head.txt   — first 40 rows of trajectory_origin.txt verbatim (OutputPoses text-format vector).
  * edge_vectors.npz   — 32 random edges: inputs and the oracle's r / J_begin / J_end (autodiff and analytic).
  * toy_graph.npz      — 120-pose ring + chords, non-identity information: LM trace and final poses (exact steps).
  * kitti00_trace.npz  — LM trace + tightly converged poses of the replay graph (oracle, exact steps).
  * g2o_00_excerpt.g2o — data file of the reference's first-iteration package (src/POSE_GRAPH/result/g2o/00.g2o):
                         the lines of vertices 0..199, "FIX 0" and every edge between them, verbatim (format vector
                         for the g2o reader: VERTEX_SE3:QUAT / FIX / EDGE_SE3:QUAT + 21 information entries).
The oracle outputs stored here pin the oracle against regressions and are what the GPU path is compared to on
the GPU box.  PARITY UNPINNED against the reference itself (see oracle/pgo_oracle.cpp header).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402
from oracle import oracle as O  # noqa: E402

ds = pgo_loader.datasets()
REF = "/root/reference/src/POSE_GRAPH_CERES_PLUS"


def rot_to_quat(R):
    """Rotation matrix -> unit quaternion xyzw (Shepperd)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0, 0, 0, 0]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def odometry_float32(poses):
    """t_be of edge (begin=i, end=i-1) = toPose3d(Tcw_i * Twc_{i-1}) with float32 4x4 matrices
    (finial.cpp:214-215, Frame.cc:165 inverse in float32)."""
    n = len(poses)
    T = np.zeros((n, 4, 4), dtype=np.float32)
    for i, p in enumerate(poses):
        q = p[3:] / np.linalg.norm(p[3:])
        T[i, :3, :3] = quat_to_rot(q).astype(np.float32)
        T[i, :3, 3] = p[:3].astype(np.float32)
        T[i, 3, 3] = 1
    out = np.zeros((n - 1, 7))
    for i in range(1, n):
        Tcw = np.linalg.inv(T[i]).astype(np.float32)
        Tcl = (Tcw @ T[i - 1]).astype(np.float32)
        out[i - 1, :3] = Tcl[:3, 3].astype(np.float64)
        out[i - 1, 3:] = rot_to_quat(Tcl[:3, :3].astype(np.float64))
    return out


def g2o_excerpt(n_vertices=200):
    src = "/root/reference/src/POSE_GRAPH/result/g2o/00.g2o"
    out = []
    for line in open(src):
        t = line.split()
        if not t:
            continue
        if t[0] == "VERTEX_SE3:QUAT" and int(t[1]) < n_vertices:
            out.append(line)
        elif t[0] == "FIX" and int(t[1]) < n_vertices:
            out.append(line)
        elif t[0] == "EDGE_SE3:QUAT" and int(t[1]) < n_vertices and int(t[2]) < n_vertices:
            out.append(line)
    open(os.path.join(HERE, "g2o_00_excerpt.g2o"), "w").write("".join(out))
    print("g2o excerpt: %d lines" % len(out))


def main():
    rng = np.random.default_rng(20260928)
    g2o_excerpt()

    # ---------------- reference data files ----------------
    ids, origin = ds.read_poses(os.path.join(REF, "result/trajectory/trajectory_origin.txt"))
    _, updated = ds.read_poses(os.path.join(REF, "result/trajectory/trajectory_update_y_not_constant.txt"))
    loops = np.loadtxt(os.path.join(REF, "result/Edges/edges_for_loop.txt"), dtype=np.int32)
    cand_path = os.path.join(REF, "config/Edge_Candidates_index.txt")
    cand_text = open(cand_path).read()
    cands = ds.read_candidates(cand_path)
    keys = sorted(k for k in cands if cands[k])
    flat = np.concatenate([np.array(cands[k], dtype=np.int32) for k in keys])
    offs = np.cumsum([0] + [len(cands[k]) for k in keys]).astype(np.int32)
    with open(os.path.join(REF, "result/trajectory/trajectory_origin.txt")) as f:
        head = "".join(f.readlines()[:40])
    open(os.path.join(HERE, "kitti00_head.txt"), "w").write(head)

    # replay graph C1: 4540 odometry + 639 loop edges
    odo = odometry_float32(origin)
    ia = np.concatenate([np.arange(1, len(origin), dtype=np.int32), loops[:, 0]])
    ib = np.concatenate([np.arange(0, len(origin) - 1, dtype=np.int32), loops[:, 1]])
    upd_n = updated.copy()
    upd_n[:, 3:] /= np.linalg.norm(upd_n[:, 3:], axis=1, keepdims=True)
    loop_meas = ds.relative_pose(upd_n[loops[:, 0]], upd_n[loops[:, 1]])  # STAND-IN measurements
    meas = np.concatenate([odo, loop_meas])
    np.savez_compressed(os.path.join(HERE, "kitti00.npz"), origin=origin, updated=updated, loops=loops,
                        cand_keys=np.array(keys, dtype=np.int32), cand_flat=flat, cand_offsets=offs,
                        cand_sha256=np.array(hashlib.sha256(cand_text.encode()).hexdigest()),
                        ia=ia, ib=ib, meas=meas)

    g = O.Graph(origin, ia, ib, meas, None)
    opt = O.default_options(max_num_iterations=1000)   # the reference's options (finial.cpp:534-536)
    p_def, s_def, tr_def = O.solve(g, opt)
    opt_t = O.default_options(max_num_iterations=1000, function_tolerance=1e-15, parameter_tolerance=1e-13)
    p_tight, s_tight, tr_tight = O.solve(g, opt_t)
    np.savez_compressed(os.path.join(HERE, "kitti00_trace.npz"), trace_default=tr_def, poses_default=p_def,
                        final_cost_default=s_def.final_cost, reason_default=s_def.reason,
                        trace_tight=tr_tight, poses_tight=p_tight, final_cost_tight=s_tight.final_cost,
                        reason_tight=s_tight.reason, initial_cost=s_def.initial_cost)
    print("kitti00 replay: initial %.6e default-stop %.9e (%d its, %s) tight %.12e (%d its, %s)" % (
        s_def.initial_cost, s_def.final_cost, s_def.num_iterations, O.REASON[s_def.reason], s_tight.final_cost,
        s_tight.num_iterations, O.REASON[s_tight.reason]))

    # ---------------- per-edge vectors ----------------
    n = 32
    pa, pb, mp = rng.normal(0, 3, (n, 3)), rng.normal(0, 3, (n, 3)), rng.normal(0, 1, (n, 3))
    qa, qb, mq = rng.normal(size=(n, 4)), rng.normal(size=(n, 4)), rng.normal(size=(n, 4))
    for q in (qa, qb, mq):
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    qa[16:] *= (1 + 1e-3 * rng.normal(size=(16, 1)))       # half of the cases are not exactly unit
    A = rng.normal(size=(n, 6, 6))
    info = A @ np.transpose(A, (0, 2, 1)) + 6 * np.eye(6)
    L = np.array([O.chol6(m) for m in info])
    L[:4] = np.eye(6)
    r_an, ja_an, jb_an, r_ad, ja_ad, jb_ad = [], [], [], [], [], []
    for i in range(n):
        r, a, b = O.edge_eval(pa[i], qa[i], pb[i], qb[i], mp[i], mq[i], L[i], "analytic")
        r_an.append(r); ja_an.append(a); jb_an.append(b)
        r, a, b = O.edge_eval(pa[i], qa[i], pb[i], qb[i], mp[i], mq[i], L[i], "autodiff")
        r_ad.append(r); ja_ad.append(a); jb_ad.append(b)
    s_vals = np.array([0.0, 0.25, 1.0, 1.0000001, 4.0, 1e6])
    rho = np.array([O.loss(1, 1.0, s) for s in s_vals])
    np.savez_compressed(os.path.join(HERE, "edge_vectors.npz"), pa=pa, qa=qa, pb=pb, qb=qb, mp=mp, mq=mq, L=L,
                        information=info, r_analytic=np.array(r_an), ja_analytic=np.array(ja_an),
                        jb_analytic=np.array(jb_an), r_autodiff=np.array(r_ad), ja_autodiff=np.array(ja_ad),
                        jb_autodiff=np.array(jb_ad), huber_s=s_vals, huber_rho=rho)

    # ---------------- toy graph ----------------
    m = 120
    th = 2 * np.pi * np.arange(m) / m
    truth = np.zeros((m, 7))
    truth[:, 0], truth[:, 1], truth[:, 2] = 10 * np.cos(th), 10 * np.sin(th), 0.5 * np.sin(3 * th)
    truth[:, 5], truth[:, 6] = np.sin((th + np.pi / 2) / 2), np.cos((th + np.pi / 2) / 2)
    tia = list(range(1, m)) + [0]
    tib = list(range(0, m - 1)) + [m - 1]
    for _ in range(60):
        a, b = rng.integers(0, m, 2)
        if a != b:
            tia.append(int(a)); tib.append(int(b))
    tia, tib = np.array(tia, dtype=np.int32), np.array(tib, dtype=np.int32)
    tmeas = ds.relative_pose(truth[tia], truth[tib])
    tmeas[:, :3] += rng.normal(0, 0.05, (len(tia), 3))
    tmeas[:, 3:] = ds.qmul(ds.qexp_half(rng.normal(0, 0.01, (len(tia), 3))), tmeas[:, 3:])
    Ai = rng.normal(size=(len(tia), 6, 6)) * 0.2
    tinfo = Ai @ np.transpose(Ai, (0, 2, 1)) + np.diag([4, 4, 4, 25, 25, 25.0])
    tL = np.array([O.chol6(x) for x in tinfo]).reshape(-1, 36)
    init = truth.copy()
    init[1:, :3] += rng.normal(0, 0.3, (m - 1, 3))
    init[1:, 3:] = ds.qmul(ds.qexp_half(rng.normal(0, 0.05, (m - 1, 3))), init[1:, 3:])
    tg = O.Graph(init, tia, tib, tmeas, tL)
    tp, tsum, ttr = O.solve(tg, O.default_options(max_num_iterations=100, function_tolerance=1e-12))
    np.savez_compressed(os.path.join(HERE, "toy_graph.npz"), poses=init, ia=tia, ib=tib, meas=tmeas, sqrt_info=tL,
                        trace=ttr, final_poses=tp, final_cost=tsum.final_cost, initial_cost=tsum.initial_cost)
    print("toy graph: %.6e -> %.9e in %d its (%s)" % (tsum.initial_cost, tsum.final_cost, tsum.num_iterations, O.REASON[tsum.reason]))


if __name__ == "__main__":
    main()
