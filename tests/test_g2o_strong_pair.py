"""CPU: pins the oracle to the SHARPEST solve result the reference holds for this path.

tests/golden/g2o_pair_strong.npz = /root/reference/src/POSE_GRAPH/result/result_before.g2o -> result/result_after.g2o
(the pair one directory ABOVE result/g2o/; same 4541 vertices / 4695 edge ids, different edge measurements; written by the
same program, test/pose_graph_try1.cpp:137-148 = save / initializeOptimization / optimize(1000) / save, edges built at
:195-215 = `EdgeSE3`, identity information, `RobustKernelHuber`, `FIX 0`).  Stored as arrays by
tests/golden/make_g2o_pair.py (data only).

Why this pair pins what result/g2o/ cannot (tests/test_g2o_pair.py: optimum fixed by "all residuals ~ 0", cost 6e-4,
vertices move 3.3 mm, Huber never active):
  * at *before* the cost is 5359.24 and 4056 of the 4695 edges sit in Huber's linear region (translation residuals of
    1.7 m median / 3.3 m max); the vertex orientations cover half turns (min |q.w| = 2.7e-4);
  * *after* lies 480 m (mean) / 931 m (max) away from *before*, at cost 2.99939 with EVERY edge carrying residual — the
    optimum is a balance of residual, Jacobian and the rotation/translation weighting, not a zero of the residuals;
  * *after* is a stationary point of THIS repo's cost (g2o chi2 == this path's cost with L = diag(1,1,1,.5,.5,.5), see
    tests/test_g2o_pair.py) to the files' print precision: ||g||_inf = 2.75e-3, and exact LM steps from *after* to tight
    convergence move the vertices 0.43 mm (mean) / 1.26 mm (max) at coordinates of ~900 m (cost -> 2.99881).  Six printed
    digits at 480 m are +-0.5 .. 5 mm per coordinate: the reference's output is reproduced to its own rounding;
  * controls on the same data: identity L (Ceres' functor without the 1/2) walks 12.9 m (mean) away from *after*, rotation
    weight x0.25 walks 17.9 m; a swapped edge direction or w-first quaternions raise the cost at *after* from 3.0 to
    5360 / 2347.
How sharply the stationary point itself is defined: the graph is a 4541-vertex chain with few closures, its softest modes
have curvature ~1e-7, so FP64 rounding of the cost (3e-16) leaves the minimiser undetermined by ~0.1-0.5 mm: a 1e-9 m
perturbation of the start moves the tightly converged end point by 0.11 mm at a cost equal to 15 digits (asserted below).
Two correct FP64 implementations (oracle, HIP path) therefore agree on the end point to ~0.5 mm and on its cost to 1e-12.
What it does NOT give: a replayable path.  From *before* the oracle's LM (Ceres 1.13 rules) descends into a different,
LOWER minimum (cost 0.548 after 300 iterations, still creeping; ~370 m from *after*): g2o's LM (other damping rule, initial
guess re-propagated by computeInitialGuess, :141-142) chose another basin of this non-convex problem.  The stationary point
is what can be checked, and is.  No xfail: the test asserts the different basin as a fact.
"""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# asserted tolerances (measured values in the module docstring)
COST_AFTER, COST_AFTER_TOL = 2.99939, 1e-5
GRAD_INF_MAX = 3e-3
STAT_MEAN_M, STAT_MAX_M = 0.6e-3, 1.5e-3
TIGHT = dict(max_num_iterations=300, function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-16)


def unit(p):
    p = p.copy()
    p[:, 3:] /= np.linalg.norm(p[:, 3:], axis=1, keepdims=True)   # g2o normalises what it reads
    return p


def rot_weight(E, w):
    return np.tile(np.diag([1.0, 1.0, 1.0, w, w, w]).reshape(-1), (E, 1))


def load_strong():
    k = np.load(os.path.join(G, "g2o_pair_strong.npz"))
    E = len(k["ia"])
    return dict(before=unit(k["before"]), after=unit(k["after"]), ia=k["ia"], ib=k["ib"], meas=unit(k["meas"]),
                L=rot_weight(E, 0.5), E=E)


def gradient_from_blocks(N, ia, ib, r, ja, jb):
    """g = sum_e J_e^T r_e over both endpoints (r, J as the evaluation returns them: robustified), pose 0 fixed."""
    g = np.zeros((N, 6))
    np.add.at(g, ia, np.einsum("eij,ei->ej", ja, r))
    np.add.at(g, ib, np.einsum("eij,ei->ej", jb, r))
    g[0] = 0
    return g


def moved(p, q):
    d = np.linalg.norm(p[:, :3] - q[:, :3], axis=1)
    return d.mean(), d.max()


@pytest.fixture(scope="module")
def sp():
    return load_strong()


def test_fixture_is_the_strong_reference_pair(sp):
    weak = np.load(os.path.join(G, "g2o_pair.npz"))
    assert sp["before"].shape == sp["after"].shape == (4541, 7) and sp["E"] == 4695
    assert np.array_equal(sp["ia"], weak["ia"]) and np.array_equal(sp["ib"], weak["ib"])
    assert not np.array_equal(sp["meas"], unit(weak["meas"]))             # not the result/g2o/ pair
    assert np.array_equal(sp["before"][0], sp["after"][0])                 # FIX 0
    mean, mx = moved(sp["after"], sp["before"])
    assert 480.0 < mean < 481.0 and 930.0 < mx < 931.5


def test_before_is_a_huber_dominated_state(O, sp):
    g = O.Graph(sp["before"], sp["ia"], sp["ib"], sp["meas"], sp["L"])
    _, r_raw, _, _ = O.evaluate(g, loss_kind=0)
    s = (r_raw ** 2).sum(1)
    assert int((s > 1.0).sum()) == 4056
    assert O.cost(g) == pytest.approx(5359.2393, rel=1e-7)
    t = np.linalg.norm(r_raw[:, :3], axis=1)
    assert 1.6 < np.median(t) < 1.8 and 3.2 < t.max() < 3.3
    assert np.abs(sp["before"][:, 6]).min() < 1e-3                       # orientations up to a half turn from the identity


def test_reference_after_is_a_stationary_point_of_this_cost(O, sp):
    g = O.Graph(sp["after"], sp["ia"], sp["ib"], sp["meas"], sp["L"])
    cost, r, ja, jb = O.evaluate(g)
    assert cost == pytest.approx(COST_AFTER, abs=COST_AFTER_TOL)
    assert int(((r ** 2).sum(1) > 1.0).sum()) == 0
    assert (np.linalg.norm(r, axis=1) > 1e-4).mean() > 0.9               # residual on (nearly) every edge: not a zero-residual optimum
    grad = gradient_from_blocks(4541, sp["ia"], sp["ib"], r, ja, jb)
    assert np.abs(grad).max() <= GRAD_INF_MAX
    p, s, _ = O.solve(g, O.default_options(**TIGHT))
    assert s.termination_type == 0
    mean, mx = moved(p, sp["after"])
    assert mean <= STAT_MEAN_M and mx <= STAT_MAX_M, (mean, mx)
    assert s.final_cost == pytest.approx(2.99881, abs=2e-5)
    assert np.array_equal(p[0], sp["after"][0])


def test_the_stationary_point_is_defined_to_tenths_of_a_millimetre_in_fp64(O, sp):
    """Why GPU-vs-oracle end points are compared at 1e-3 m and their costs at 1e-12 (tests/test_gpu_g2o_strong_pair.py)."""
    opt = O.default_options(**TIGHT)
    p0, s0, _ = O.solve(O.Graph(sp["after"], sp["ia"], sp["ib"], sp["meas"], sp["L"]), opt)
    nudged = sp["after"].copy()
    nudged[1:, :3] += 1e-9 * np.random.default_rng(1).standard_normal((4540, 3))
    p1, s1, _ = O.solve(O.Graph(nudged, sp["ia"], sp["ib"], sp["meas"], sp["L"]), opt)
    assert s1.final_cost == pytest.approx(s0.final_cost, rel=1e-13)
    d = np.abs(p1[:, :3] - p0[:, :3]).max()
    assert 1e-5 < d < 1e-3, d
    p2, s2, _ = O.solve(O.Graph(p0, sp["ia"], sp["ib"], sp["meas"], sp["L"]), opt)   # restarting at the end point: nothing moves
    assert np.abs(p2[:, :3] - p0[:, :3]).max() < 1e-9


def test_wrong_weighting_or_conventions_leave_the_reference_after(O, sp):
    ia, ib, m, A, E = sp["ia"], sp["ib"], sp["meas"], sp["after"], sp["E"]
    opt = O.default_options(**TIGHT)
    p_id, _, _ = O.solve(O.Graph(A, ia, ib, m, None), opt)                # Ceres' functor with identity information
    assert moved(p_id, A)[0] > 5.0
    p_q, _, _ = O.solve(O.Graph(A, ia, ib, m, rot_weight(E, 0.25)), opt)  # rotation weight halved again
    assert moved(p_q, A)[0] > 5.0
    assert O.cost(O.Graph(A, ib, ia, m, sp["L"])) > 1e3                   # edge direction swapped
    wxyz = m.copy()
    wxyz[:, 3:] = m[:, [6, 3, 4, 5]]
    assert O.cost(O.Graph(A, ia, ib, wxyz, sp["L"])) > 1e3                # quaternion stored w first


def test_the_path_from_before_is_not_replayable_and_why(O, sp):
    """Ceres-1.13 LM from *before* reaches a LOWER cost than the reference's *after* in another basin: stated, not hidden."""
    g = O.Graph(sp["before"], sp["ia"], sp["ib"], sp["meas"], sp["L"])
    p, s, tr = O.solve(g, O.default_options(max_num_iterations=300))
    assert s.initial_cost == pytest.approx(5359.2393, rel=1e-7)
    assert s.final_cost < 1.0 < COST_AFTER
    assert moved(p, sp["after"])[0] > 100.0
