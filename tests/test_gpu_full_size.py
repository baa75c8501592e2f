"""-m gpu: BASELINE.json's full-size configurations through size-independent properties (the oracle still finishes
in seconds for one evaluation sweep, not for a solve): cost/gradient agreement, descent, determinism, constant pose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(ds):
    return ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)   # 100 k poses / 1 M edges


def test_c4_cost_gradient_and_lm_descent(gpu, ds, O, big):
    g = big
    assert (g.N, g.E) == (100000, 1000000)
    prob, poses = gpu.problem_from_graph(g)
    cost, _, _, _, grad = prob.evaluate(residuals=False, jacobians=False, gradient=True)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    assert cost == pytest.approx(O.cost(og), rel=1e-11)
    # gradient against the oracle on a random subset of edges' endpoints is expensive; use linearity instead:
    # the directional derivative of the cost along d equals grad . d (central difference through Plus)
    rng = np.random.default_rng(0)
    d = rng.normal(size=(g.N, 6)) * 1e-6
    d[0] = 0
    base = poses.copy()
    prob.plus(d)
    cp = prob.evaluate(False, False, False)[0]
    poses[:] = base
    prob.plus(-d)
    cm = prob.evaluate(False, False, False)[0]
    poses[:] = base
    assert (cp - cm) / 2 == pytest.approx(float((grad * d).sum()), rel=2e-5)
    assert not grad[0].any()                                   # constant first pose
    # a few LM iterations: monotone descent of accepted steps, bitwise reproducible, pose 0 untouched
    opt = gpu.SolverOptions(max_num_iterations=4, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    s1 = gpu.solve(opt, prob)
    acc = s1.iterations["cost"][s1.iterations["step_is_successful"] == 1]
    assert len(acc) >= 2 and np.all(np.diff(acc) < 0)
    assert np.array_equal(poses[0], g.poses[0])
    prob2, poses2 = gpu.problem_from_graph(g)
    s2 = gpu.solve(opt, prob2)
    assert np.array_equal(poses, poses2) and s1.final_cost == s2.final_cost


def test_c4_symmetric_form_session_is_tied_to_the_oracle(gpu, ds, O, big):
    """The DEFAULT path of BASELINE configs[3] on one GPU once a solve can repay the form (max_num_iterations >= 64): the normal
    equations live in the symmetric tile form only (k_linearize_lean writes it, k_spmv_sym reads it).  Tied to the oracle at full
    size by what one oracle sweep can check: the cost the session reports for its final poses is the oracle's cost of those
    poses (1e-11), the accepted costs of its trace are the oracle's costs of ... the same statement at the start, and the gradient
    the form's linearisation produced at the final point satisfies the directional-derivative identity."""
    g = big
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=64, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
    assert s.cg_form == 2 and s.sym_form == 1 and len(s.iterations) >= 20          # above the universal stream's size limit: the one-launch pipelined CG iteration on the form (r06: k_pipe_cg_sym)
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    assert s.initial_cost == pytest.approx(O.cost(og), rel=1e-11)
    assert s.final_cost == pytest.approx(O.cost(og, poses), rel=1e-11)          # one oracle evaluation of the FINAL poses
    assert s.final_cost < 0.05 * s.initial_cost
    assert np.array_equal(poses[0], g.poses[0])
    # the gradient at the final point (fresh evaluation through the incidence-slot kernels) against central differences of the cost
    final = poses.copy()
    cost, _, _, _, grad = prob.evaluate(residuals=False, jacobians=False, gradient=True)
    assert cost == pytest.approx(s.final_cost, rel=1e-11)
    d = np.random.default_rng(1).normal(size=(g.N, 6)) * 1e-6
    d[0] = 0
    prob.plus(d)
    cp = prob.evaluate(False, False, False)[0]
    poses[:] = final
    prob.plus(-d)
    cm = prob.evaluate(False, False, False)[0]
    poses[:] = final
    assert (cp - cm) / 2 == pytest.approx(float((grad * d).sum()), rel=1e-4, abs=1e-9 * abs(cost))
    # and the session is reproducible bit for bit
    prob2, poses2 = gpu.problem_from_graph(g)
    s2 = gpu.solve(gpu.SolverOptions(max_num_iterations=64, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob2)
    assert s2.final_cost == s.final_cost and np.array_equal(poses2, final)


def test_c5_sphere_layers(gpu, ds, O):
    """SURVEY C5 shape: 10 sphere2500-style layers, 25 k poses / 250 k edges (single GPU here)."""
    g = ds.sphere_layers(n_spheres=10, rings=50, per_ring=50, n_edges=250000, seed=20260931)
    assert (g.N, g.E) == (25000, 250000)
    prob, poses = gpu.problem_from_graph(g)
    cost = prob.evaluate(False, False, False)[0]
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    assert cost == pytest.approx(O.cost(og), rel=1e-11)
    s = gpu.solve(gpu.SolverOptions(max_num_iterations=6, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=6, linear_solver=1, pcg_cluster=2))
    n = min(len(s.iterations), len(otr))
    assert list(s.iterations["linear_solver_iterations"][:n]) == [int(v) for v in otr[:n, 7]]
    assert np.allclose(s.iterations["cost"][:n], otr[:n, 1], rtol=1e-7)


def test_c2_exact_request_is_served_either_way_with_the_same_answer(gpu, ds, monkeypatch):
    """C2 (10 k poses / 40 k edges) with the reference's linear solver setting and the multifrontal solver switched off
    (PGO_FRONT=0; the default path for this graph is tests/test_gpu_front.py): the enumerated factorisation is admitted but
    costly, so each LM iteration is served by it or by PCG to 1e-13 (linear_solver_used = 3).  Size-independent properties: the
    LM trace equals the one obtained with the factorisation switched off (PCG to 1e-13 alone) and the one with the
    factorisation alone, and a repeated run is bit-identical."""
    monkeypatch.setenv("PGO_FRONT", "0")
    g = ds.manhattan_se3(10000, 40000)
    opt = lambda: gpu.SolverOptions(max_num_iterations=10, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)

    def run():
        prob, poses = gpu.problem_from_graph(g)
        return gpu.solve(opt(), prob), poses

    s, x = run()
    assert s.linear_solver_used == 3 and s.factor_nnz_blocks > 100000
    s2, x2 = run()
    assert s2.final_cost == s.final_cost and np.array_equal(x, x2)
    monkeypatch.setenv("PGO_DIRECT_MAX_STEPS", "1000000")          # factorisation for every iteration
    sd, xd = run()
    assert sd.linear_solver_used == 0
    monkeypatch.delenv("PGO_DIRECT_MAX_STEPS")
    monkeypatch.setenv("PGO_NO_DIRECT", "1")                       # PCG to 1e-13 for every iteration
    sp, xp = run()
    assert sp.linear_solver_used == 2
    for other in (sd, sp):
        n = len(s.iterations)
        assert len(other.iterations) == n
        assert list(other.iterations["step_is_successful"]) == list(s.iterations["step_is_successful"])
        assert np.allclose(other.iterations["cost"], s.iterations["cost"], rtol=1e-7)
    assert np.abs(xd - x).max() < 1e-5 and np.abs(xp - x).max() < 1e-5


def test_c4_sharded_eight_ways_matches_one_rank(gpu, ds, big):
    """BASELINE.json configs[3] as it is specified: 100 k poses / 1 M edges SHARDED 8 WAYS.  Eight RCCL ranks need eight GPUs;
    on this box the eight ranks are virtual (loopback transport: one host thread and one problem per rank, the all-gathers are
    event-ordered device copies — the same ownership rule, kernels, exchange points and replicated decisions as over RCCL).
    Every rank must take the decisions of the single-rank solve: same accept / reject sequence, same CG iteration count in
    every LM iteration, costs to 1e-9, poses to 1e-7, and all eight ranks bit-identical to each other."""
    import threading
    g = big
    world = 8
    opt = dict(max_num_iterations=4, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    prob, poses = gpu.problem_from_graph(g)
    ref = gpu.solve(gpu.SolverOptions(**opt), prob)
    del prob
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            p, x = gpu.problem_from_graph(g)
            p.comm_init_loopback(group, rank)
            out[rank] = (gpu.solve(gpu.SolverOptions(**opt), p), x)
        except Exception as e:      # a failing rank would leave the others waiting: surface it
            errs.append(e)

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errs, errs
    assert all(o is not None for o in out), "a virtual rank did not finish"
    gpu.loopback_destroy(group)
    for s, x in out:
        assert s.sym_form == 1 and s.cg_form == 2           # (r06) every rank keeps the symmetric form of its rows: k_pipe_cg_sym, k_linearize_lean
        assert list(s.iterations["step_is_successful"]) == list(ref.iterations["step_is_successful"])
        assert list(s.iterations["linear_solver_iterations"]) == list(ref.iterations["linear_solver_iterations"])
        assert np.allclose(s.iterations["cost"], ref.iterations["cost"], rtol=1e-9)
        assert np.abs(x - poses).max() < 1e-7
        assert np.array_equal(x, out[0][1])


def test_c2_pcg_with_a_tight_forcing_term_reaches_the_exact_paths_cost(gpu, ds):
    """north_star: "results match the reference ... on final pose error" — the reference solves every LM step exactly
    (finial.cpp:534-536).  Ceres' default forcing term eta = 0.1 makes cheap steps but, run to its own stop from dead reckoning,
    ends 11 % above the exact path's cost on BASELINE configs[1] (another basin).  With eta = 1e-5 the same PCG path reaches the
    exact path's final cost within 1e-3 (measured -1.8e-4: slightly BELOW it), in a third of the wall time."""
    g = ds.manhattan_se3(10000, 40000)
    pe, _ = gpu.problem_from_graph(g)
    ex = gpu.solve(gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), pe)
    pp, _ = gpu.problem_from_graph(g)
    pc = gpu.solve(gpu.SolverOptions(max_num_iterations=3000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=1e-5,
                                     max_linear_solver_iterations=3000), pp)
    pl, _ = gpu.problem_from_graph(g)
    loose = gpu.solve(gpu.SolverOptions(max_num_iterations=3000, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), pl)
    assert ex.termination_type == gpu.CONVERGENCE and pc.termination_type == gpu.CONVERGENCE
    assert pc.final_cost <= ex.final_cost * (1.0 + 1e-3), (pc.final_cost, ex.final_cost)
    assert loose.final_cost > ex.final_cost * 1.05           # the documented weakness of eta = 0.1 (if this ever fails: update DESIGN.md)
    assert pc.total_time_in_seconds < ex.total_time_in_seconds


def test_c4_pcg_forcing_terms_agree_once_tight(gpu, ds):
    """BASELINE configs[3] on one GPU (no factorisation of this size exists to compare with): eta = 1e-4 and eta = 1e-5 end within
    1e-2 of each other, eta = 0.01 ends 7 % above.  (How close the two tight runs end is itself sensitive to the last bits of the
    products: with the r05 tile partition they ended 2e-4 apart after 52 / 63 LM iterations; with tiles made of whole 2-pose clusters
    (r06) — the same matrix, another summation order — the eta = 1e-4 run meets the function tolerance at iteration 31, 4.6e-3 above the
    eta = 1e-5 run, which moved by 6e-6.  Measured in one gpurun call with the unit switched: EXPERIMENTS.md r06.)"""
    g = ds.manhattan_se3(100000, 1000000, seed=20260930, loop_radius=3.0)
    out = {}
    for eta in (1e-2, 1e-4, 1e-5):
        prob, _ = gpu.problem_from_graph(g)
        out[eta] = gpu.solve(gpu.SolverOptions(max_num_iterations=400, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, eta=eta,
                                               max_linear_solver_iterations=3000), prob)
    assert abs(out[1e-4].final_cost / out[1e-5].final_cost - 1.0) <= 1e-2
    assert out[1e-2].final_cost > 1.03 * out[1e-5].final_cost
