"""Pins the C++ oracle (oracle/pgo_oracle.cpp) against an independent numpy/scipy restatement with
finite differences, and its two Jacobian routes (AutoDiff chain vs closed form) against each other.
CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import np_ref


def _rq(rng, n=None):
    q = rng.normal(size=(4,) if n is None else (n, 4))
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def test_residual_and_jacobians_match_numpy_fd(O):
    rng = np.random.default_rng(1)
    for t in range(25):
        pa, pb = np.concatenate([rng.normal(size=3), _rq(rng)]), np.concatenate([rng.normal(size=3), _rq(rng)])
        meas = np.concatenate([rng.normal(size=3), _rq(rng)])
        A = rng.normal(size=(6, 6))
        L = O.chol6(A @ A.T + 6 * np.eye(6)) if t % 3 else np.eye(6)
        r, Ja, Jb = O.edge_eval(pa[:3], pa[3:], pb[:3], pb[3:], meas[:3], meas[3:], L, "analytic")
        assert np.allclose(r, np_ref.residual(pa, pb, meas, L), rtol=0, atol=1e-12)
        Fa, Fb = np_ref.fd_jacobians(pa, pb, meas, L)
        assert np.abs(Ja - Fa).max() < 5e-8 * max(1, np.abs(Fa).max())
        assert np.abs(Jb - Fb).max() < 5e-8 * max(1, np.abs(Fb).max())


def test_autodiff_chain_equals_closed_form_also_off_the_sphere(O):
    rng = np.random.default_rng(2)
    for t in range(200):
        qa, qb = _rq(rng) * (1 + 1e-2 * rng.normal()), _rq(rng) * (1 + 1e-2 * rng.normal())
        args = (rng.normal(size=3), qa, rng.normal(size=3), qb, rng.normal(size=3), _rq(rng))
        A = rng.normal(size=(6, 6))
        L = O.chol6(A @ A.T + 6 * np.eye(6))
        r1, a1, b1 = O.edge_eval(*args, L, "analytic")
        r2, a2, b2 = O.edge_eval(*args, L, "autodiff")
        assert np.abs(r1 - r2).max() < 1e-12 and np.abs(a1 - a2).max() < 1e-11 and np.abs(b1 - b2).max() < 1e-11


def test_switchable_constraint_loss_is_the_eliminated_switch(O):
    """PGO_LOSS_SWITCHABLE: rho(s) = min_w [w^2 s + Phi (1 - w)^2] with w* = Phi / (Phi + s); rho' = w*^2 = d rho / d s."""
    for phi in (0.5, 5.0):
        for s in (0.0, 0.3, 2.0, 50.0):
            rho = O.loss(5, phi, s)
            w = np.linspace(0.0, 1.0, 200001)
            assert rho[0] == pytest.approx((w * w * s + phi * (1 - w) ** 2).min(), abs=1e-9)
            assert rho[1] == pytest.approx((phi / (phi + s)) ** 2, rel=1e-14)
            h = 1e-6 * max(1.0, s)
            assert rho[1] == pytest.approx((O.loss(5, phi, s + h)[0] - O.loss(5, phi, max(0.0, s - h))[0]) / (h + min(h, s)), rel=1e-5)
            assert rho[2] < 0.0


def test_huber_and_plus(O):
    for s in [0.0, 0.5, 1.0, 1.5, 100.0]:
        rho = O.loss(1, 1.0, s)
        r0, r1 = np_ref.huber(s)
        assert rho[0] == pytest.approx(r0) and rho[1] == pytest.approx(r1)
        assert rho[2] == (0.0 if s <= 1.0 else pytest.approx(-r1 / (2 * s)))
    assert list(O.loss(0, 1.0, 9.0)) == [9.0, 1.0, 0.0]
    rng = np.random.default_rng(3)
    for _ in range(20):
        q, d = _rq(rng), rng.normal(0, 0.4, 3)
        pose = np.concatenate([np.zeros(3), q])
        assert np.allclose(O.quat_plus(q, d), np_ref.plus(pose, np.concatenate([np.zeros(3), d]))[3:], atol=1e-15)
        assert abs(np.linalg.norm(O.quat_plus(q, d)) - 1) < 1e-14
    q = _rq(rng)
    assert np.array_equal(O.quat_plus(q, np.zeros(3)), q)  # |delta| == 0 leaves q bitwise


def _small_graph(ds, n=40, e=110, seed=5):
    g = ds.manhattan_se3(n, e, seed=seed, loop_radius=4.0, min_gap=3)
    return g


def test_normal_equations_and_cost_match_numpy(O, ds):
    g = _small_graph(ds)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    Ls = g.sqrt_info.reshape(-1, 6, 6)
    assert O.cost(og) == pytest.approx(np_ref.cost(g.poses, g.ia, g.ib, g.meas, Ls), rel=1e-12)
    c, H, grad = O.normal_equations_dense(og)
    free = list(range(1, g.N))
    Hn, gn = np_ref.normal_equations(g.poses, g.ia, g.ib, g.meas, Ls, free)
    Hf, gf = H[6:, 6:], grad[6:]
    assert np.abs(Hf - Hn).max() < 1e-6 * np.abs(Hn).max()
    assert np.abs(gf - gn).max() < 1e-6 * np.abs(gn).max()
    # constant pose 0: decoupled unit diagonal, zero gradient
    assert np.array_equal(H[:6, :6], np.eye(6)) and not H[:6, 6:].any() and not grad[:6].any()


def test_exact_solver_and_pcg_match_scipy(O, ds):
    g = _small_graph(ds, 80, 260, seed=6)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    c, H, grad = O.normal_equations_dense(og)
    rng = np.random.default_rng(0)
    d2 = rng.uniform(0.01, 1.0, size=H.shape[0])
    b = rng.normal(size=H.shape[0])
    xs = spla.spsolve(sp.csc_matrix(H + np.diag(d2)), b)
    x, it = O.linear_solve(og, d2, b, linear_solver=0)
    assert it == 0 and np.abs(x - xs).max() < 1e-10 * np.abs(xs).max()
    # PCG run to a tiny Q-tolerance converges to the same solution
    x2, it2 = O.linear_solve(og, d2, b, linear_solver=1, q_tol=1e-14, max_it=5000)
    assert it2 > 5 and np.abs(x2 - xs).max() < 1e-7 * np.abs(xs).max()


def _numpy_lm(g, max_it, loss=True):
    """Dense restatement of SURVEY.md Appendix A.6 with FD Jacobians (tiny graphs only)."""
    poses = g.poses.copy()
    Ls = g.sqrt_info.reshape(-1, 6, 6)
    free = list(range(1, g.N))
    H, gr = np_ref.normal_equations(poses, g.ia, g.ib, g.meas, Ls, free, loss)
    S = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    radius, dec = 1e4, 2.0
    cost = np_ref.cost(poses, g.ia, g.ib, g.meas, Ls, loss)
    costs, accepted = [cost], [True]
    for _ in range(max_it):
        Hs, gs = H * S[:, None] * S[None, :], gr * S
        D2 = np.clip(np.diag(Hs), 1e-6, 1e32) / radius
        y = np.linalg.solve(Hs + np.diag(D2), gs)
        step = -y
        model = -step @ gs - 0.5 * step @ Hs @ step
        delta = step * S
        cand = poses.copy()
        for i, v in enumerate(free):
            cand[v] = np_ref.plus(poses[v], delta[6 * i:6 * i + 6])
        c_new = np_ref.cost(cand, g.ia, g.ib, g.meas, Ls, loss)
        if abs(cost - c_new) <= 1e-6 * cost:
            break
        rho = (cost - c_new) / model
        if rho > 1e-3:
            poses, cost = cand, c_new
            H, gr = np_ref.normal_equations(poses, g.ia, g.ib, g.meas, Ls, free, loss)
            radius = min(1e16, radius / max(1 / 3.0, 1 - (2 * rho - 1) ** 3))
            dec = 2.0
            accepted.append(True)
        else:
            radius /= dec
            dec *= 2
            accepted.append(False)
        costs.append(c_new)
    return poses, costs, accepted


@pytest.mark.parametrize("loss", [True, False])
def test_lm_loop_matches_dense_numpy_lm(O, ds, loss):
    g = _small_graph(ds, 30, 80, seed=8)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    poses, s, tr = O.solve(og, O.default_options(max_num_iterations=50, loss_kind=1 if loss else 0))
    np_poses, costs, accepted = _numpy_lm(g, 50, loss)
    n = min(len(costs), len(tr))
    assert n >= 4
    assert [bool(v) for v in tr[:n, 8]] == accepted[:n]
    assert np.allclose(tr[:n, 1], costs[:n], rtol=2e-6)
    assert np.abs(poses - np_poses).max() < 1e-4


def test_options_defaults_are_the_ceres_1_13_values(O):
    o = O.default_options()
    assert (o.max_num_iterations, o.max_linear_solver_iterations, o.jacobi_scaling) == (50, 500, 1)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius) == (1e4, 1e16, 1e-32)
    assert (o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal, o.eta) == (1e-3, 1e-6, 1e32, 0.1)
