"""Independent numpy/scipy restatement of the path's math, used ONLY to pin the C++ oracle
(tests/test_oracle_numpy.py).  It is written differently on purpose: rotation matrices via scipy,
Jacobians by central differences through the retraction, dense normal equations."""
import numpy as np
from scipy.spatial.transform import Rotation


def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def plus(pose, d):
    """p += dp ; q <- [sin|t| t/|t|, cos|t|] * q  (half-angle vector, left multiplication)."""
    out = pose.copy()
    out[:3] += d[:3]
    n = np.linalg.norm(d[3:])
    if n > 0:
        dq = np.concatenate([np.sin(n) * d[3:] / n, [np.cos(n)]])
        out[3:] = qmul(dq, pose[3:])
    return out


def residual(pose_a, pose_b, meas, L):
    """PoseGraph3dError.h:21-54 for unit quaternions, through rotation matrices."""
    Ra = Rotation.from_quat(pose_a[3:]).as_matrix()
    p_ab = Ra.T @ (pose_b[:3] - pose_a[:3])
    qa_inv = pose_a[3:] * np.array([-1, -1, -1, 1.0])
    q_ab = qmul(qa_inv, pose_b[3:])
    dq = qmul(meas[3:], q_ab * np.array([-1, -1, -1, 1.0]))
    e = np.concatenate([p_ab - meas[:3], 2.0 * dq[:3]])
    return L @ e


def huber(s, a=1.0):
    if s > a * a:
        r = np.sqrt(s)
        return 2 * a * r - a * a, a / r
    return s, 1.0


def fd_jacobians(pose_a, pose_b, meas, L, h=1e-6):
    Ja, Jb = np.zeros((6, 6)), np.zeros((6, 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        Ja[:, k] = (residual(plus(pose_a, d), pose_b, meas, L) - residual(plus(pose_a, -d), pose_b, meas, L)) / (2 * h)
        Jb[:, k] = (residual(pose_a, plus(pose_b, d), meas, L) - residual(pose_a, plus(pose_b, -d), meas, L)) / (2 * h)
    return Ja, Jb


def cost(poses, ia, ib, meas, Ls, loss=True):
    c = 0.0
    for e in range(len(ia)):
        r = residual(poses[ia[e]], poses[ib[e]], meas[e], Ls[e])
        s = r @ r
        c += 0.5 * (huber(s)[0] if loss else s)
    return c


def normal_equations(poses, ia, ib, meas, Ls, free, loss=True, h=1e-6):
    """Dense H = J'J, g = J'r over the free poses (list of pose indices), FD Jacobians, Huber sqrt(rho') scaling."""
    idx = {v: i for i, v in enumerate(free)}
    m = 6 * len(free)
    H, g = np.zeros((m, m)), np.zeros(m)
    for e in range(len(ia)):
        a, b = ia[e], ib[e]
        r = residual(poses[a], poses[b], meas[e], Ls[e])
        Ja, Jb = fd_jacobians(poses[a], poses[b], meas[e], Ls[e], h)
        w = np.sqrt(huber(r @ r)[1]) if loss else 1.0
        r, Ja, Jb = w * r, w * Ja, w * Jb
        blocks = [(a, Ja), (b, Jb)]
        for v, J in blocks:
            if v in idx:
                i = 6 * idx[v]
                g[i:i + 6] += J.T @ r
                for u, K in blocks:
                    if u in idx:
                        j = 6 * idx[u]
                        H[i:i + 6, j:j + 6] += J.T @ K
    return H, g
