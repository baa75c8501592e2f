"""Pose-graph construction bookkeeping around the solve: the front-end rules of
pose_graph_ceres_plus_finial.cpp with the vision parts (ORB matching, PnP-RANSAC) replaced by RECORDED measurements.

Host logic only (numpy); the edges it produces feed Problem.add_se3_between / datasets.PoseGraphData.

Reference behaviour restated here:
  * one odometry edge per frame, id_begin = current, id_end = previous, t_be = T_cur<-prev = Tcw(cur) * Twc(prev),
    identity information (finial.cpp:206-224);
  * a candidate pair more than one frame apart is examined only when it has > 280 ORB matches (finial.cpp:226), and
    accepted when PnP reports > 100 inliers and normofTransform < 0.7 (finial.cpp:234; norm = finial.cpp:486-489);
  * at most one loop edge per current frame, and a candidate that already obtained a loop edge when it was the current
    frame is skipped (finial.cpp:238, 288-289; the flag set on the candidate is set on a copy and is lost);
  * accepted pairs more than 100 frames apart are listed in edges_for_loop.txt as "cur prev" (finial.cpp:285-286);
  * Tcl (4x4, float32 in the reference) -> Pose3d by Converter::toPose3d (converter.cc:221-234): p = translation,
    q = Eigen quaternion of the rotation block;
  * after the solve the y coordinate of every pose is overwritten with the input trajectory's y (finial.cpp:145-156).
"""
import math

import numpy as np

MATCH_THRESHOLD = 280      # finial.cpp:226
INLIER_THRESHOLD = 100     # finial.cpp:234
NORM_THRESHOLD = 0.7       # finial.cpp:234
LOOP_LIST_GAP = 100        # finial.cpp:285


def _norm3(v):
    """sqrt((x*x + y*y) + z*z) with every operation rounded separately, in this order: the C-ABI twin (csrc/pgo_edges.hip)
    states the same sequence, so both sides agree bit for bit (np.linalg.norm may go through a scaled BLAS routine)."""
    x, y, z = (float(c) for c in np.asarray(v, dtype=np.float64).reshape(3))
    return math.sqrt((x * x + y * y) + z * z)


def norm_of_transform(rvec, tvec):
    """finial.cpp:486-489."""
    r = _norm3(rvec)
    t = _norm3(tvec)
    return abs(min(r, 2.0 * math.pi - r)) + abs(t)


def rodrigues(rvec):
    """Rotation vector -> 3x3 rotation matrix (cv::Rodrigues, finial.cpp:256)."""
    r = [float(c) for c in np.asarray(rvec, dtype=np.float64).reshape(3)]
    th = _norm3(r)
    if th < 2.2204460492503131e-16:
        return np.eye(3)
    k = [r[0] / th, r[1] / th, r[2] / th]
    K = [[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]]
    c, s = math.cos(th), math.sin(th)
    omc = 1.0 - c
    return np.array([[(c * (1.0 if i == j else 0.0) + omc * (k[i] * k[j])) + s * K[i][j] for j in range(3)] for i in range(3)])


def quaternion_from_matrix(R):
    """Eigen::Quaterniond(Matrix3d) (converter.cc:150-155): Shepperd's branch on the trace, result [x, y, z, w]."""
    m = np.asarray(R, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0.0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def to_pose3d(T):
    """Converter::toPose3d (converter.cc:221-234): 4x4 (or 3x4) transform -> [px py pz qx qy qz qw]."""
    T = np.asarray(T, dtype=np.float64)
    return np.concatenate([T[:3, 3], quaternion_from_matrix(T[:3, :3])])


def relative_transform(Twc_cur, Twc_prev):
    """Tcl = Tcw(cur) * Twc(prev) (finial.cpp:213).  Inputs: 4x4 camera-to-world transforms.  Plain Python floats, every
    product and sum rounded separately in index order (no BLAS): the same arithmetic as the GPU kernel k_odometry_edges."""
    C = [[float(x) for x in row] for row in np.asarray(Twc_cur, dtype=np.float64)]
    P = [[float(x) for x in row] for row in np.asarray(Twc_prev, dtype=np.float64)]
    Tcw = [[0.0] * 4 for _ in range(4)]
    for i in range(3):
        for j in range(3):
            Tcw[i][j] = C[j][i]
        Tcw[i][3] = -((C[0][i] * C[0][3] + C[1][i] * C[1][3]) + C[2][i] * C[2][3])
    Tcw[3][3] = 1.0
    T = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            T[i, j] = ((Tcw[i][0] * P[0][j] + Tcw[i][1] * P[1][j]) + Tcw[i][2] * P[2][j]) + Tcw[i][3] * P[3][j]
    return T


class LoopEdgeBuilder:
    """Replays checkForPoseGraph / checkFrame (finial.cpp:162-293) over recorded front-end results.

    observations(cur, prev) -> None or dict(nmatches=int, inliers=int, rvec=(3,), tvec=(3,)) stands in for
    ORBmatcher::MatcheTwoFrames + motionEstimate.  Frames must be added in id order."""

    def __init__(self, match_threshold=MATCH_THRESHOLD, inlier_threshold=INLIER_THRESHOLD, norm_threshold=NORM_THRESHOLD,
                 loop_list_gap=LOOP_LIST_GAP, float32_transforms=True):
        self.match_threshold, self.inlier_threshold = match_threshold, inlier_threshold
        self.norm_threshold, self.loop_list_gap = norm_threshold, loop_list_gap
        self.float32 = float32_transforms
        self.Twc = {}
        self.have_loop_edge = {}
        self.ia, self.ib, self.meas = [], [], []
        self.loop_list = []          # rows of edges_for_loop.txt
        self.log = []

    def _add_edge(self, cur, prev, T):
        if self.float32:
            T = np.asarray(T, dtype=np.float32)      # the reference keeps Tcl in CV_32F before toPose3d
        self.ia.append(cur)
        self.ib.append(prev)
        self.meas.append(to_pose3d(T))

    def add_frame(self, frame_id, Twc, candidates=(), observations=None):
        """One pass of the main loop (finial.cpp:86-110) for frame `frame_id`: registers the vertex pose, then checks the
        candidate frames (the ids Edge_Candidates_index.txt lists for this frame, in file order)."""
        Twc = np.asarray(Twc, dtype=np.float64)
        if Twc.shape == (3, 4):
            Twc = np.vstack([Twc, [0.0, 0.0, 0.0, 1.0]])
        self.Twc[frame_id] = Twc
        have = False
        for prev in candidates:
            if prev not in self.Twc or prev == frame_id:
                continue                                   # the reference would index out of range; recorded data never does
            if frame_id - prev == 1:
                self._add_edge(frame_id, prev, relative_transform(Twc, self.Twc[prev]))
                continue
            if frame_id - prev <= 1:
                continue
            obs = observations(frame_id, prev) if observations else None
            if obs is None or not obs["nmatches"] > self.match_threshold:
                continue
            norm = norm_of_transform(obs["rvec"], obs["tvec"])
            if not (obs["inliers"] > self.inlier_threshold and norm < self.norm_threshold):
                continue
            if have or self.have_loop_edge.get(prev, False):
                continue
            T = np.eye(4)
            T[:3, :3] = rodrigues(obs["rvec"])
            T[:3, 3] = np.asarray(obs["tvec"], dtype=np.float64).reshape(3)
            self._add_edge(frame_id, prev, T)
            if frame_id - prev > self.loop_list_gap:
                self.loop_list.append((frame_id, prev))
            have = True
            self.log.append((frame_id, prev, obs["inliers"], norm))
        self.have_loop_edge[frame_id] = have

    def vertex_poses(self):
        ids = sorted(self.Twc)
        return ids, np.array([to_pose3d(self.Twc[i]) for i in ids])

    def edges(self):
        return (np.array(self.ia, dtype=np.int32), np.array(self.ib, dtype=np.int32),
                np.array(self.meas, dtype=np.float64).reshape(-1, 7))

    def format_loop_list(self):
        return "".join("%d %d\n" % ab for ab in self.loop_list)


def overwrite_y(poses, input_xyz):
    """finial.cpp:145-156: after the solve, pose y is replaced by the input trajectory's (float32) y."""
    out = np.array(poses, dtype=np.float64, copy=True)
    out[:, 1] = np.asarray(input_xyz, dtype=np.float32)[: out.shape[0], 1].astype(np.float64)
    return out
