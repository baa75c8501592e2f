"""Pose-graph inputs either side of the hot path: synthetic generators for the BASELINE.json
configs and the reference's text formats.

Reference formats followed (REF = /root/reference/src/POSE_GRAPH_CERES_PLUS):
  * OutputPoses text .................. REF/test/pose_graph_ceres_plus_finial.cpp:547-567
  * Edge_Candidates_index.txt reader .. REF/include/ReadEdges.h:9-47
  * candidate generation rule ......... REF/test/generate_edges_from_trajectory_origion.cpp:58-111
  * edge direction / odometry rule .... finial.cpp:206-224 (id_begin = current, id_end = previous,
                                        t_be = T_cur<-prev), SURVEY.md Appendix C
  * g2o text .......................... src/POSE_GRAPH/result/g2o/00.g2o (VERTEX_SE3:QUAT / EDGE_SE3:QUAT)
Quaternions are Hamilton, stored x,y,z,w (Eigen coeffs order, REF/include/types.h:15-20).
"""
import numpy as np

# ------------------------------------------------------------------------------------------------
# quaternion helpers (vectorised, xyzw)
# ------------------------------------------------------------------------------------------------


def qmul(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qconj(q):
    q = np.asarray(q, dtype=np.float64)
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def qrot(q, v):
    """Rotate v by unit quaternion q (Eigen's v + 2w(u x v) + 2 u x (u x v))."""
    q = np.asarray(q, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    u, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def qexp_half(delta):
    """[sin|d| d/|d|, cos|d|]: the quaternion EigenQuaternionParameterization::Plus left-multiplies."""
    delta = np.asarray(delta, dtype=np.float64)
    n = np.linalg.norm(delta, axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(n > 0, np.sin(n) / n, 1.0)
    return np.concatenate([s * delta, np.cos(n)], axis=-1)


def relative_pose(pose_a, pose_b):
    """Measurement t_ab that makes the reference residual zero: p = R_a^T (p_b - p_a), q = q_a^* q_b
    (REF/include/PoseGraph3dError.h:32-36).  pose arrays are (...,7)."""
    pa, qa = pose_a[..., :3], pose_a[..., 3:]
    pb, qb = pose_b[..., :3], pose_b[..., 3:]
    qai = qconj(qa)
    return np.concatenate([qrot(qai, pb - pa), qmul(qai, qb)], axis=-1)


def compose_from_measurement(pose_b, meas):
    """Given pose_b (id_end) and t_ab, return pose_a (id_begin) with zero residual:
    q_a = q_b q^*, p_a = p_b - R(q_a) p."""
    pb, qb = pose_b[..., :3], pose_b[..., 3:]
    qa = qmul(qb, qconj(meas[..., 3:]))
    pa = pb - qrot(qa, meas[..., :3])
    return np.concatenate([pa, qa], axis=-1)


# ------------------------------------------------------------------------------------------------
# containers
# ------------------------------------------------------------------------------------------------


class PoseGraphData:
    """poses (N,7) initial guess; ia/ib id_begin/id_end (int32); meas (E,7); sqrt_info (E,36) or None
    (None = identity information, as the reference always uses: finial.cpp:217-218,276-277)."""

    def __init__(self, poses, ia, ib, meas, sqrt_info=None, truth=None, name=""):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64)
        self.ia = np.ascontiguousarray(ia, dtype=np.int32)
        self.ib = np.ascontiguousarray(ib, dtype=np.int32)
        self.meas = np.ascontiguousarray(meas, dtype=np.float64)
        self.sqrt_info = None if sqrt_info is None else np.ascontiguousarray(sqrt_info, dtype=np.float64)
        self.truth = truth
        self.name = name

    @property
    def N(self):
        return self.poses.shape[0]

    @property
    def E(self):
        return self.ia.shape[0]


def _noisy_measurements(truth, ia, ib, rng, sigma_t, sigma_r):
    m = relative_pose(truth[ia], truth[ib])
    m[:, :3] += rng.normal(0.0, sigma_t, size=(len(ia), 3))
    m[:, 3:] = qmul(qexp_half(rng.normal(0.0, sigma_r, size=(len(ia), 3))), m[:, 3:])
    return m


def _dead_reckon(first_pose, meas_odo):
    """poses[i] from poses[i-1] and the odometry edge (a=i, b=i-1)."""
    n = meas_odo.shape[0] + 1
    out = np.zeros((n, 7))
    out[0] = first_pose
    # plain-Python scalar loop: 1e5 poses take about a second
    px, py, pz, qx, qy, qz, qw = [float(v) for v in first_pose]
    M = meas_odo.tolist()
    for i in range(1, n):
        mx, my, mz, nx, ny, nz, nw = M[i - 1]
        # q_a = q_b * conj(m_q)
        cx, cy, cz, cw = -nx, -ny, -nz, nw
        ax = qw * cx + qx * cw + qy * cz - qz * cy
        ay = qw * cy + qy * cw + qz * cx - qx * cz
        az = qw * cz + qz * cw + qx * cy - qy * cx
        aw = qw * cw - qx * cx - qy * cy - qz * cz
        # p_a = p_b - R(q_a) m_p
        ux, uy, uz = 2 * (ay * mz - az * my), 2 * (az * mx - ax * mz), 2 * (ax * my - ay * mx)
        rx = mx + aw * ux + (ay * uz - az * uy)
        ry = my + aw * uy + (az * ux - ax * uz)
        rz = mz + aw * uz + (ax * uy - ay * ux)
        px, py, pz = px - rx, py - ry, pz - rz
        qx, qy, qz, qw = ax, ay, az, aw
        out[i] = (px, py, pz, qx, qy, qz, qw)
    return out


def _loop_pairs(xyz, count, rng, radius, min_gap):
    from scipy.spatial import cKDTree
    n = len(xyz)
    possible = max(0, n - min_gap - 1) * max(0, n - min_gap) // 2      # pairs more than min_gap apart, at any distance
    if count > possible:
        raise ValueError("cannot place %d loop edges: %d poses admit only %d pairs more than %d ids apart" % (count, n, possible, min_gap))
    if count <= 0:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), radius
    tree = cKDTree(xyz)
    r = radius
    while True:
        pairs = tree.query_pairs(r, output_type="ndarray")
        if len(pairs):
            lo = np.minimum(pairs[:, 0], pairs[:, 1])
            hi = np.maximum(pairs[:, 0], pairs[:, 1])
            keep = (hi - lo) > min_gap
            lo, hi = lo[keep], hi[keep]
        else:
            lo = hi = np.zeros(0, dtype=np.int64)
        if len(lo) >= count:
            break
        r *= 1.5
    sel = rng.choice(len(lo), size=count, replace=False)
    sel.sort()
    return hi[sel].astype(np.int32), lo[sel].astype(np.int32), r


def manhattan_se3(n_poses=10000, n_edges=40000, seed=20260928, sigma_t=0.05, sigma_r=0.01,
                  loop_radius=3.0, min_gap=20, identity_information=False):
    """BASELINE.json configs[1] / SURVEY.md §8d C2: Manhattan-world SE(3) walk, 1 m steps, yaw +-90deg
    w.p. 0.3 (and pitch +-90deg w.p. 0.05 to leave the plane), N(0,0.01 rad) attitude jitter; loop edges
    between poses within `loop_radius` and id gap > min_gap; diag information 1/sigma^2."""
    rng = np.random.default_rng(seed)
    n = n_poses
    truth = np.zeros((n, 7))
    truth[0, 6] = 1.0
    turn = rng.random(n)
    sign = rng.choice([-1.0, 1.0], size=n)
    jitter = rng.normal(0.0, 0.01, size=(n, 3))
    q = np.array([0.0, 0.0, 0.0, 1.0])
    p = np.zeros(3)
    h = np.sqrt(0.5)
    for i in range(1, n):
        dq = qexp_half(0.5 * jitter[i])
        if turn[i] < 0.30:
            dq = qmul(np.array([0.0, 0.0, sign[i] * h, h]), dq)      # yaw +-90
        elif turn[i] < 0.35:
            dq = qmul(np.array([0.0, sign[i] * h, 0.0, h]), dq)      # pitch +-90
        q = qmul(q, dq)
        q /= np.linalg.norm(q)
        p = p + qrot(q, np.array([1.0, 0.0, 0.0]))
        truth[i, :3] = p
        truth[i, 3:] = q
    n_odo = n - 1
    n_loop = n_edges - n_odo
    ia_o = np.arange(1, n, dtype=np.int32)
    ib_o = np.arange(0, n - 1, dtype=np.int32)
    ia_l, ib_l, _ = _loop_pairs(truth[:, :3], n_loop, rng, loop_radius, min_gap)
    ia = np.concatenate([ia_o, ia_l])
    ib = np.concatenate([ib_o, ib_l])
    meas = _noisy_measurements(truth, ia, ib, rng, sigma_t, sigma_r)
    init = _dead_reckon(truth[0], meas[:n_odo])
    sqrt_info = None
    if not identity_information:
        L = np.diag([1.0 / sigma_t] * 3 + [1.0 / sigma_r] * 3).reshape(1, 36)
        sqrt_info = np.repeat(L, len(ia), axis=0)
    return PoseGraphData(init, ia, ib, meas, sqrt_info, truth=truth, name="manhattan_se3_%d_%d" % (n, len(ia)))


class PoseLandmarkData:
    """A pose / landmark problem: `graph` = the pose part (odometry between-factors), points (M,3) initial
    guesses, observations obs_pose / obs_point (indices into poses / points), obs_z (K,3) = the point in the observing pose's
    frame, obs_sqrt_info3 (K,9).  as_pose_graph(): the same problem as ONE pose graph — every point a node with a constant
    identity quaternion, every observation a between-factor whose information has no rotation part — which is what both the
    oracle and the product solve (nodes: poses first, then points; returns the graph and the constant mask)."""

    def __init__(self, graph, points, obs_pose, obs_point, obs_z, obs_sqrt_info3, truth_points=None):
        self.graph = graph
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        self.obs_pose = np.ascontiguousarray(obs_pose, dtype=np.int32)
        self.obs_point = np.ascontiguousarray(obs_point, dtype=np.int32)
        self.obs_z = np.ascontiguousarray(obs_z, dtype=np.float64)
        self.obs_sqrt_info3 = np.ascontiguousarray(obs_sqrt_info3, dtype=np.float64)
        self.truth_points = truth_points

    def as_pose_graph(self):
        g = self.graph
        N, M, K = g.N, self.points.shape[0], len(self.obs_pose)
        poses = np.zeros((N + M, 7))
        poses[:N] = g.poses
        poses[N:, :3] = self.points
        poses[N:, 6] = 1.0
        meas = np.zeros((K, 7))
        meas[:, :3] = self.obs_z
        meas[:, 6] = 1.0
        L = np.zeros((K, 6, 6))
        L[:, :3, :3] = self.obs_sqrt_info3.reshape(K, 3, 3)
        si_pose = g.sqrt_info if g.sqrt_info is not None else np.repeat(np.eye(6).reshape(1, 36), g.E, axis=0)
        full = PoseGraphData(poses, np.concatenate([g.ia, self.obs_pose]), np.concatenate([g.ib, N + self.obs_point]),
                             np.concatenate([g.meas, meas]), np.concatenate([si_pose, L.reshape(K, 36)]))
        cmask = np.zeros(N + M, dtype=np.uint8)
        cmask[0] = 3
        cmask[N:] = 2
        return full, cmask


def pose_landmark_toy(n_poses=150, n_points=1500, seed=20260932, sigma_t=0.05, sigma_r=0.01, sigma_z=0.03, view_radius=4.0, max_views=8):
    """A trajectory (the Manhattan walk of manhattan_se3, odometry factors only) through a cloud of 3-D points, every point
    observed from up to `max_views` poses within `view_radius`; observation = the point in the pose's frame + N(0, sigma_z)."""
    from scipy.spatial import cKDTree
    g = manhattan_se3(n_poses, n_poses - 1, seed=seed, sigma_t=sigma_t, sigma_r=sigma_r)
    rng = np.random.default_rng(seed + 1)
    truth = g.truth
    anchor = rng.integers(0, n_poses, size=n_points)
    pts = truth[anchor, :3] + rng.uniform(-2.0, 2.0, size=(n_points, 3))
    tree = cKDTree(truth[:, :3])
    op, ol = [], []
    for j, nb in enumerate(tree.query_ball_point(pts, view_radius)):
        nb = sorted(nb)
        if len(nb) > max_views:
            nb = sorted(rng.choice(nb, size=max_views, replace=False).tolist())
        if not nb:
            nb = [int(anchor[j])]
        op += nb
        ol += [j] * len(nb)
    op, ol = np.asarray(op, dtype=np.int32), np.asarray(ol, dtype=np.int32)
    z = qrot(qconj(truth[op, 3:]), pts[ol] - truth[op, :3]) + rng.normal(0.0, sigma_z, size=(len(op), 3))
    # initial points: from the first observation and the dead-reckoned pose of its observer
    first = np.full(n_points, -1)
    for k in range(len(op) - 1, -1, -1):
        first[ol[k]] = k
    init = g.poses[op[first], :3] + qrot(g.poses[op[first], 3:], z[first])
    L3 = np.repeat((np.eye(3) / sigma_z).reshape(1, 9), len(op), axis=0)
    return PoseLandmarkData(g, init, op, ol, z, L3, truth_points=pts)


def sphere_layers(n_spheres=10, rings=50, per_ring=50, radius=50.0, seed=20260931, chord_radius=8.0,
                  n_edges=None, sigma_t=0.05, sigma_r=0.01):
    """sphere2500-style layouts chained; ring/meridian neighbours + random chords."""
    rng = np.random.default_rng(seed)
    pts = []
    for s in range(n_spheres):
        centre = np.array([2.2 * radius * s, 0.0, 0.0])
        for r in range(rings):
            phi = np.pi * (r + 0.5) / rings
            for k in range(per_ring):
                th = 2 * np.pi * k / per_ring
                pos = centre + radius * np.array([np.sin(phi) * np.cos(th), np.sin(phi) * np.sin(th), np.cos(phi)])
                # heading tangent to the ring
                yaw = th + np.pi / 2
                qz = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
                pts.append(np.concatenate([pos, qz]))
    truth = np.array(pts)
    n = truth.shape[0]
    ia = [np.arange(1, n, dtype=np.int32)]
    ib = [np.arange(0, n - 1, dtype=np.int32)]
    idx = np.arange(n)
    # meridian neighbours: same k on the next ring
    m = idx[per_ring:]
    same = (m // (rings * per_ring)) == ((m - per_ring) // (rings * per_ring))
    ia.append(m[same].astype(np.int32))
    ib.append((m[same] - per_ring).astype(np.int32))
    ia = np.concatenate(ia)
    ib = np.concatenate(ib)
    if n_edges is None:
        n_edges = 10 * n
    extra = n_edges - len(ia)
    if extra > 0:
        la, lb, _ = _loop_pairs(truth[:, :3], extra, rng, chord_radius, per_ring + 1)
        ia = np.concatenate([ia, la])
        ib = np.concatenate([ib, lb])
    meas = _noisy_measurements(truth, ia, ib, rng, sigma_t, sigma_r)
    init = _dead_reckon(truth[0], meas[: n - 1])
    L = np.diag([1.0 / sigma_t] * 3 + [1.0 / sigma_r] * 3).reshape(1, 36)
    return PoseGraphData(init, ia, ib, meas, np.repeat(L, len(ia), axis=0), truth=truth, name="sphere_layers_%d_%d" % (n, len(ia)))


def graph_from_candidates(poses, candidates, seed=20260929, sigma_t=0.05, sigma_r=0.01, init="trajectory"):
    """SURVEY.md §8d C3 (KITTI-00 dense): every id in Edge_Candidates_index.txt becomes an edge
    (id_begin = line key, id_end = candidate id); measurements synthesised from the given trajectory
    taken as ground truth + noise.  Initial guess: the given trajectory itself (init="trajectory", what the
    reference starts from: the poses of trajectory_origin.txt) or dead reckoning of the noisy odometry edges."""
    init_mode = init
    rng = np.random.default_rng(seed)
    truth = np.array(poses, dtype=np.float64)
    truth[:, 3:] /= np.linalg.norm(truth[:, 3:], axis=1, keepdims=True)
    ia, ib = [], []
    for k in sorted(candidates):
        for c in candidates[k]:
            if 0 <= c < len(truth) and k < len(truth):
                ia.append(k)
                ib.append(c)
    ia = np.array(ia, dtype=np.int32)
    ib = np.array(ib, dtype=np.int32)
    meas = _noisy_measurements(truth, ia, ib, rng, sigma_t, sigma_r)
    # odometry edges are the (k, k-1) entries: first candidate of every line
    odo = np.where(ia - ib == 1)[0]
    order = odo[np.argsort(ia[odo])]
    init = truth.copy()
    if init_mode == "dead_reckoning" and len(order) == len(truth) - 1:
        init = _dead_reckon(truth[0], meas[order])
    return PoseGraphData(init, ia, ib, meas, None, truth=truth, name="candidates_%d_%d" % (len(truth), len(ia)))


# ------------------------------------------------------------------------------------------------
# reference text formats
# ------------------------------------------------------------------------------------------------


def _g6(v):
    """C++ default ostream formatting of a double (== printf %g, precision 6)."""
    return "%g" % v


def format_pose_line(pid, pose):
    """One OutputPoses row (finial.cpp:561-564): `id p.transpose() qx qy qz qw`; Eigen's default
    IOFormat right-aligns the three coefficients of p to their common maximum width."""
    ps = [_g6(pose[0]), _g6(pose[1]), _g6(pose[2])]
    w = max(len(s) for s in ps)
    ps = " ".join(s.rjust(w) for s in ps)
    return "%d %s %s %s %s %s\n" % (pid, ps, _g6(pose[3]), _g6(pose[4]), _g6(pose[5]), _g6(pose[6]))


def write_poses(path, poses, ids=None):
    ids = range(len(poses)) if ids is None else ids
    with open(path, "w") as f:
        for i, p in zip(ids, poses):
            f.write(format_pose_line(i, p))


def read_poses(path):
    """Rows `id x y z qx qy qz qw` -> (ids int array, poses (N,7))."""
    a = np.loadtxt(path, dtype=np.float64, ndmin=2)
    return a[:, 0].astype(np.int64), np.ascontiguousarray(a[:, 1:8])


def read_candidates(path):
    """ReadEdges.h:9-47: key = 1-based line number, first token of each line skipped; a trailing
    blank line yields an empty extra entry (kept, as the reference does)."""
    out = {}
    with open(path) as f:
        text = f.read()
    lines = text.split("\n")
    for i, s in enumerate(lines, start=1):
        toks = s.split()
        out[i] = [int(t) for t in toks[1:]]
    return out


def generate_candidates(xyz, search_radius=6.0, gap=100):
    """generate_edges_from_trajectory_origion.cpp:58-111: for frame id>=1 write `id id-1` followed by
    every i < id-gap whose float32 squared distance to frame id is <= radius^2.  Returns dict id->list."""
    p = np.asarray(xyz, dtype=np.float32)
    r2 = np.float32(search_radius) * np.float32(search_radius)
    out = {}
    for k in range(1, len(p)):
        row = [k - 1]
        hi = k - gap
        if hi > 0:
            d = p[:hi] - p[k]
            d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            row.extend(np.nonzero(d2 <= r2)[0].tolist())
        out[k] = row
    return out


def format_candidates(cands):
    return "".join("%d %s \n" % (k, " ".join(str(c) for c in cands[k])) for k in sorted(cands))


def read_g2o(path):
    """VERTEX_SE3:QUAT / EDGE_SE3:QUAT / FIX.  g2o's edge (i,j) measures pose of j in frame i, i.e.
    id_begin=i, id_end=j in the reference's Edge3d convention.  Information: 21 upper-triangular
    entries -> full 6x6 -> sqrt_info = lower Cholesky factor L (finial.cpp:508)."""
    ids, poses, ia, ib, meas, infos, fixed = [], [], [], [], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "VERTEX_SE3:QUAT":
                ids.append(int(t[1]))
                poses.append([float(x) for x in t[2:9]])
            elif t[0] == "EDGE_SE3:QUAT":
                ia.append(int(t[1]))
                ib.append(int(t[2]))
                meas.append([float(x) for x in t[3:10]])
                u = [float(x) for x in t[10:31]]
                M = np.zeros((6, 6))
                k = 0
                for r in range(6):
                    for c in range(r, 6):
                        M[r, c] = M[c, r] = u[k]
                        k += 1
                infos.append(M)
            elif t[0] == "FIX":
                fixed.append(int(t[1]))
    ids = np.array(ids)
    remap = {int(v): i for i, v in enumerate(ids)}
    ia = np.array([remap[i] for i in ia], dtype=np.int32)
    ib = np.array([remap[i] for i in ib], dtype=np.int32)
    infos = np.array(infos)
    sqrt_info = None
    if len(infos) and not np.allclose(infos, np.eye(6)[None]):
        sqrt_info = np.linalg.cholesky(infos).reshape(-1, 36)
    g = PoseGraphData(np.array(poses), ia, ib, np.array(meas), sqrt_info, name=str(path))
    g.fixed = [remap[i] for i in fixed]
    g.ids = ids
    return g


def write_g2o(path, g, fixed=(0,), exact=False):
    """exact=True prints 17 significant digits (lossless round trip) instead of the 6 g2o tools print."""
    fmt = (lambda v: "%.17g" % v) if exact else _g6
    _g = fmt
    with open(path, "w") as f:
        for i, p in enumerate(g.poses):
            f.write("VERTEX_SE3:QUAT %d %s \n" % (i, " ".join(_g(v) for v in p)))
            if i in fixed:
                f.write("FIX %d\n" % i)
        for e in range(g.E):
            if g.sqrt_info is None:
                info = np.eye(6)
            else:
                L = g.sqrt_info[e].reshape(6, 6)
                info = L @ L.T
            up = [info[r, c] for r in range(6) for c in range(r, 6)]
            f.write("EDGE_SE3:QUAT %d %d %s %s \n" % (g.ia[e], g.ib[e], " ".join(_g(v) for v in g.meas[e]),
                                                    " ".join(_g(v) for v in up)))
