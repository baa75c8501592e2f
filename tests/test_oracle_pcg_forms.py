"""CPU: the oracle's two statements of the preconditioned CG (oracle/pgo_oracle.cpp pcg_solve): form 0 = Ceres 1.13's
ConjugateGradientsSolver statement by statement, form 1 = the pipelined recurrences (Ghysels & Vanroose 2014) the product's
one-launch CG iteration computes (csrc/pgo_uni_fused.h).  Same Krylov iterates in exact arithmetic, same stop rules on the same
quantities: on well-posed systems they stop after the same number of iterations at solutions that agree to rounding, and a numpy
restatement of the pipelined recurrences (written from the paper's algorithm, independent of the C++) reproduces form 1."""
import numpy as np
import pytest


def _system(O, ds, seed, cluster):
    g = ds.manhattan_se3(300, 1100, seed=seed)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, H, grad = O.normal_equations_dense(og)
    rng = np.random.default_rng(seed)
    d2 = np.full(6 * g.N, 1e-1) * (1 + rng.random(6 * g.N))
    b = rng.normal(size=6 * g.N)
    b[:6] = 0
    return og, H + np.diag(d2), d2, b


def _numpy_pipelined(A, b, cluster, q_tol, max_it):
    n = len(b)
    dim = 6 * cluster
    Minv = np.zeros_like(A)
    for s in range(0, n, dim):
        e = min(n, s + dim)
        Minv[s:e, s:e] = np.linalg.inv(A[s:e, s:e])
    x = np.zeros(n); r = b.copy(); u = Minv @ r; w = A @ u
    z = np.zeros(n); q = np.zeros(n); s_ = np.zeros(n); p = np.zeros(n)
    gamma_prev = alpha_prev = q_prev = 0.0
    cnt = 0
    while True:
        m = Minv @ w
        gamma, delta, Q1 = r @ u, w @ u, -(x @ (b + r))
        if cnt > 0 and (cnt * (Q1 - q_prev) / Q1 < q_tol or cnt >= max_it):
            break
        beta = gamma / gamma_prev if cnt > 0 else 0.0
        alpha = gamma / (delta - beta * gamma / alpha_prev) if cnt > 0 else gamma / delta
        nn = A @ m
        z = nn + beta * z; q = m + beta * q; s_ = w + beta * s_; p = u + beta * p
        x = x + alpha * p; r = r - alpha * s_; u = u - alpha * q; w = w - alpha * z
        gamma_prev, alpha_prev, q_prev = gamma, alpha, Q1
        cnt += 1
    return x, cnt


@pytest.mark.parametrize("cluster", [1, 2])
@pytest.mark.parametrize("q_tol", [0.1, 1e-3])
def test_pipelined_and_standard_forms_stop_together_at_the_same_solution(O, ds, cluster, q_tol):
    og, A, d2, b = _system(O, ds, 5, cluster)
    code = 1 if cluster == 1 else 100 + cluster
    x0, it0 = O.linear_solve(og, d2, b, linear_solver=code, q_tol=q_tol, max_it=500)
    x1, it1 = O.linear_solve(og, d2, b, linear_solver=1000 + code, q_tol=q_tol, max_it=500)
    assert it0 == it1 and 3 < it0 < 500
    assert np.abs(x0 - x1).max() <= 1e-9 * np.abs(x0).max()
    assert not np.array_equal(x0, x1)                       # (two different sequences of roundings, not one code path)
    xn, itn = _numpy_pipelined(A, b, cluster, q_tol, 500)
    assert itn == it1
    assert np.abs(xn - x1).max() <= 1e-8 * np.abs(x1).max()     # (numpy: explicit block inverses; the oracle: Cholesky solves)


def test_lm_traces_of_the_two_forms_agree(O, ds):
    g = ds.manhattan_se3(600, 2400, seed=9)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    res = [O.solve(og, O.default_options(max_num_iterations=15, linear_solver=1, pcg_cluster=2, pcg_form=f)) for f in (0, 1)]
    (p0, s0, t0), (p1, s1, t1) = res
    assert s0.num_iterations == s1.num_iterations
    assert [int(x) for x in t0[:, 8]] == [int(x) for x in t1[:, 8]]            # decisions
    assert [int(x) for x in t0[:, 7]] == [int(x) for x in t1[:, 7]]            # CG iterations per LM iteration
    assert np.allclose(t0[:, 1], t1[:, 1], rtol=1e-5)       # (rejected candidates behind CG runs of hundreds of iterations: 6e-7 measured)
    assert np.abs(p0 - p1).max() < 1e-6
