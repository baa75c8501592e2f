"""MotionEstimate reprojection problem.  CPU: the oracle restatement
(oracle_reproj_*: functor of REF/include/MotionEstimate.h:34-91 through a 7-wide Jet + the Plus Jacobian, Ceres LM) against
central differences and against the known solution.  GPU: the batched kernel (pgo_reproj_solve_batch) against the oracle
problem by problem — same iteration counts and stopping reasons, costs to 1e-9, parameters to 1e-8."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

INTR = np.array([718.856, 718.856, 607.1928, 185.2157])     # KITTI 00 (REF/config/default.yaml Camera.*)


def project(q, t, P):
    X = Rotation.from_quat(q).apply(P) + t
    return np.c_[INTR[0] * X[:, 0] / X[:, 2] + INTR[2], INTR[1] * X[:, 1] / X[:, 2] + INTR[3]]


def make_problem(rng, n, outliers=0.05, noise=0.5):
    P = np.c_[rng.uniform(-12, 12, n), rng.uniform(-3, 3, n), rng.uniform(4, 45, n)]
    q_true = Rotation.from_rotvec(rng.normal(0, 0.03, 3)).as_quat()
    t_true = rng.normal(0, 0.5, 3)
    obs = project(q_true, t_true, P) + rng.normal(0, noise, (n, 2))
    k = int(outliers * n)
    if k:
        obs[:k] += rng.normal(0, 40, (k, 2))
    return P, obs, q_true, t_true


def test_oracle_jacobian_and_recovery(O):
    rng = np.random.default_rng(0)
    P, obs, q_true, t_true = make_problem(rng, 300)
    q0 = Rotation.from_rotvec([0.01, 0.0, -0.02]).as_quat() * 1.0003      # not exactly unit: the chain must still match
    t0 = np.array([0.1, -0.2, 0.05])
    r, J = O.reproj_eval(P, obs, INTR, q0, t0)

    def res(q, t):       # Eigen's unnormalised rotation formula, as the functor evaluates it
        u, w = q[:3], q[3]
        uv = 2 * np.cross(u, P)
        X = P + w * uv + np.cross(u, uv) + t
        return np.c_[INTR[0] * X[:, 0] / X[:, 2] + INTR[2], INTR[1] * X[:, 1] / X[:, 2] + INTR[3]] - obs

    assert np.abs(r - res(q0, t0)).max() < 1e-10
    h = 1e-6
    for c in range(6):
        d = np.zeros(6)
        d[c] = h
        fd = (res(O.quat_plus(q0, d[:3]), t0 + d[3:]) - res(O.quat_plus(q0, -d[:3]), t0 - d[3:])) / (2 * h)
        assert np.abs(J[:, :, c] - fd).max() <= 1e-6 * np.abs(J).max()
    # the reference's setting: rotation constant (here the true one), translation from zero, Huber(1)
    q, t, s, tr = O.reproj_solve(P, obs, INTR, q_true, np.zeros(3), cmask=2)
    assert s.termination_type == 0 and np.abs(t - t_true).max() < 0.02 and np.array_equal(q, q_true)
    assert np.all(np.diff(tr[tr[:, 8] == 1, 1]) < 0)           # accepted steps descend
    # both blocks free
    q2, t2, s2, _ = O.reproj_solve(P, obs, INTR, np.array([0, 0, 0, 1.0]), np.zeros(3), cmask=0)
    assert s2.termination_type == 0 and s2.final_cost <= s.final_cost * 1.001
    assert np.abs(Rotation.from_quat(q2).as_rotvec() - Rotation.from_quat(q_true).as_rotvec()).max() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["t_only", "q_and_t", "trivial_loss"])
def test_gpu_batch_matches_oracle(gpu, O, mode):
    rng = np.random.default_rng(11)
    sizes = [1, 2, 63, 64, 65, 200, 333, 1000, 7, 128]
    probs = [make_problem(rng, n) for n in sizes]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    pts = np.concatenate([p[0] for p in probs])
    obs = np.concatenate([p[1] for p in probs])
    nP = len(sizes)
    q = np.zeros((nP, 4))
    t = np.zeros((nP, 3))
    for k, (P, ob, q_true, t_true) in enumerate(probs):
        q[k] = q_true if mode != "q_and_t" else Rotation.from_rotvec(Rotation.from_quat(q_true).as_rotvec() + 0.01).as_quat()
    cmask = 0 if mode == "q_and_t" else 2
    loss = 0 if mode == "trivial_loss" else 1
    opt = gpu.ReprojOptions(q_constant=1 if cmask & 2 else 0, loss_kind=loss)
    q0, t0 = q.copy(), t.copy()
    summ = gpu.reproj_solve_batch(ptr, pts, obs, INTR, q, t, opt)
    for k, (P, ob, q_true, t_true) in enumerate(probs):
        oq, ot, osum, otr = O.reproj_solve(P, ob, INTR, q0[k], t0[k], cmask=cmask,
                                           options=O.default_options(max_num_iterations=1000, loss_kind=loss))
        assert summ["num_points"][k] == sizes[k]
        if sizes[k] < 3 and mode == "q_and_t":
            continue      # under-determined: both sides wander along the null space, nothing to compare
        assert summ["num_iterations"][k] == osum.num_iterations, (k, sizes[k])
        assert summ["termination_type"][k] == osum.termination_type and summ["reason"][k] == osum.reason
        assert summ["initial_cost"][k] == pytest.approx(osum.initial_cost, rel=1e-12, abs=1e-12)
        assert summ["final_cost"][k] == pytest.approx(osum.final_cost, rel=1e-9, abs=1e-12)
        assert np.abs(t[k] - ot).max() < 1e-8 and np.abs(q[k] - oq).max() < 1e-9
        if cmask & 2:
            assert np.array_equal(q[k], q0[k])


@pytest.mark.gpu
def test_gpu_batch_edge_cases(gpu):
    # no problems, a problem without points, argument checks
    assert len(gpu.reproj_solve_batch(np.array([0]), np.zeros((0, 3)), np.zeros((0, 2)), INTR, np.zeros((0, 4)), np.zeros((0, 3)))) == 0
    q = np.array([[0, 0, 0, 1.0]])
    t = np.array([[1.0, 2.0, 3.0]])
    s = gpu.reproj_solve_batch(np.array([0, 0]), np.zeros((0, 3)), np.zeros((0, 2)), INTR, q, t)
    assert s["num_points"][0] == 0 and s["final_cost"][0] == 0.0 and np.array_equal(t, [[1.0, 2.0, 3.0]])
    with pytest.raises(gpu.PgoError):
        gpu.reproj_solve_batch(np.array([0, 0]), np.zeros((0, 3)), np.zeros((0, 2)), INTR, q, t, gpu.ReprojOptions(loss_a=-1.0))
