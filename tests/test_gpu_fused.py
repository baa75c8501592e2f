"""-m gpu: the universal stream in its fused form (r05, csrc/pgo_uni_fused.h: one kernel symbol, one launch per CG iteration, the
pipelined recurrences of Ghysels & Vanroose) against
  * the oracle's restatement of the same recurrences (oracle/pgo_oracle.cpp pcg_solve form 1): same decisions, same CG iteration
    counts in every LM iteration, costs to 1e-7;
  * the two-kernel stream (standard CG, Ceres' ConjugateGradientsSolver statement by statement): same decisions and counts;
  * itself: stepping with budgets / pauses / resets equals one solve bit for bit; two runs are bit-identical;
and the launch trace (include/pgo.h pgo_solver_trace_*) obeys the stream's grammar."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["iteration", "step_is_successful", "linear_solver_iterations", "cost", "cost_change", "gradient_max_norm",
          "step_norm", "relative_decrease", "trust_region_radius"]


def _graph(ds, kind):
    if kind == "diag":          # BASELINE configs[1]'s information (diag(1 / sigma^2)), a tenth of its size
        return ds.manhattan_se3(1000, 4000, seed=3)
    g = ds.manhattan_se3(1000, 4000, seed=4)
    if kind == "identity":
        return ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, None)
    rng = np.random.default_rng(11)       # "full": position / rotation coupling in the information (36-entry slots, INFO 1)
    L = np.zeros((g.E, 6, 6))
    for e in range(g.E):
        A = np.tril(rng.normal(size=(6, 6)) * 0.3)
        A[np.diag_indices(6)] = 1.0 + rng.random(6)
        L[e] = A
    return ds.PoseGraphData(g.poses, g.ia, g.ib, g.meas, L.reshape(g.E, 36))


def _solve(gpu, g, form, **kw):
    prob, poses = gpu.problem_from_graph(g)
    opt = dict(max_num_iterations=20, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_form=form)
    opt.update(kw)
    s = gpu.solve(gpu.SolverOptions(**opt), prob)
    return s, poses


@pytest.mark.parametrize("kind", ["diag", "identity", "full"])
@pytest.mark.parametrize("cluster", [1, 2])
def test_fused_stream_matches_the_oracles_pipelined_cg(gpu, ds, O, kind, cluster, monkeypatch):
    monkeypatch.setenv("PGO_BLOCK", "256")     # (a 1000-pose graph would get 64-slot work-groups: some pose pairs do not fit them)
    g = _graph(ds, kind)
    s, poses = _solve(gpu, g, 2, pcg_cluster_poses=cluster)
    assert s.cg_form == 3                                     # the fused stream really ran
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    op, osum, otr = O.solve(og, O.default_options(max_num_iterations=20, linear_solver=1, pcg_cluster=cluster, pcg_form=1))
    n = len(otr)
    assert len(s.iterations) == n
    assert list(s.iterations["step_is_successful"]) == [int(x) for x in otr[:, 8]]
    assert list(s.iterations["linear_solver_iterations"]) == [int(x) for x in otr[:, 7]]
    assert np.allclose(s.iterations["cost"], otr[:, 1], rtol=1e-7)
    assert np.allclose(s.iterations["trust_region_radius"], otr[:, 6], rtol=1e-6)     # (a function of the cost ratios: 5e-8 measured)
    assert s.final_cost == pytest.approx(osum.final_cost, rel=1e-7)
    assert np.abs(poses - op).max() < 1e-5
    assert max(s.iterations["linear_solver_iterations"]) > 20       # long CG runs are part of what agrees


@pytest.mark.parametrize("loss", ["trivial", "huber", "cauchy"])
def test_fused_and_two_kernel_streams_agree(gpu, ds, loss, monkeypatch):
    """Standard and pipelined CG are the same Krylov iterates in exact arithmetic: same decisions, same CG counts, costs to 1e-8."""
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    res = {}
    for form in (1, 2):
        prob, poses = gpu.problem_from_graph(g, loss={"trivial": gpu.TRIVIAL, "huber": gpu.HUBER, "cauchy": gpu.CAUCHY}[loss], loss_a=1.0)
        s = gpu.solve(gpu.SolverOptions(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2,
                                        pcg_form=form), prob)
        res[form] = (s, poses)
    a, b = res[1][0], res[2][0]
    assert a.cg_form == 0 and b.cg_form == 3
    assert list(a.iterations["step_is_successful"]) == list(b.iterations["step_is_successful"])
    assert list(a.iterations["linear_solver_iterations"]) == list(b.iterations["linear_solver_iterations"])
    assert np.allclose(a.iterations["cost"], b.iterations["cost"], rtol=1e-8)
    assert np.abs(res[1][1] - res[2][1]).max() < 1e-6


def test_requests_the_fused_stream_does_not_serve_keep_the_two_kernel_stream(gpu, ds, monkeypatch):
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(1000, 4000, seed=3)
    assert _solve(gpu, g, 0, pcg_cluster_poses=2)[0].cg_form == 4            # the library's choice where it applies: a one-launch form (the resident one: tests/test_gpu_resident.py) ...
    assert _solve(gpu, g, 0, pcg_cluster_poses=2, eta=1e-3)[0].cg_form == 0  # ... which excludes tight forcing terms (long CG runs: Ceres' refreshed CG)
    assert _solve(gpu, g, 2, pcg_cluster_poses=2, eta=1e-3)[0].cg_form == 3  # unless asked for
    assert _solve(gpu, g, 1, pcg_cluster_poses=2)[0].cg_form == 0            # the caller asked for Ceres' recurrences
    assert _solve(gpu, g, 2, pcg_cluster_poses=4)[0].cg_form == 0            # 24 x 24 Jacobi blocks
    # an exact request answered by PCG runs the CG to a relative residual: standard form, whatever was asked
    s, _ = _solve(gpu, g, 2, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, max_num_iterations=3)
    assert s.cg_form == 0


def test_fused_stepping_pauses_and_resets_equal_one_solve(gpu, ds, monkeypatch):
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(1500, 6000, seed=21)
    opt = dict(max_num_iterations=40, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=2)
    ref, pref = _solve(gpu, g, 2, **{k: v for k, v in opt.items() if k != "pcg_form"})
    prob, poses = gpu.problem_from_graph(g)
    prob.solver_begin(gpu.SolverOptions(**opt))
    prob.solver_step(7)
    prob.solver_reset()                       # back to the initial poses: the 7 steps leave no trace
    done, total = False, 0
    for n in (1, 2, 1, 5, 3, 100):
        if done:
            break
        ran, done = prob.solver_step(n)
        total += ran
    s = prob.solver_end()
    assert s.cg_form == 3 and done
    assert len(s.iterations) == len(ref.iterations)
    for f in FIELDS:
        assert np.array_equal(s.iterations[f], ref.iterations[f]), f
    assert s.final_cost == ref.final_cost and s.message == ref.message
    assert np.array_equal(poses, pref)
    again, pagain = _solve(gpu, g, 2, **{k: v for k, v in opt.items() if k != "pcg_form"})      # run to run: the same bits
    for f in FIELDS:
        assert np.array_equal(again.iterations[f], ref.iterations[f]), f
    assert np.array_equal(pagain, pref)


def test_launch_trace_obeys_the_streams_grammar(gpu, ds, monkeypatch):
    """HEAD W0 CG* (the last CG launch multiplies A x) TAIL [LIN behind an accepted step] HEAD ...: one launch per CG iteration."""
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(1500, 6000, seed=21)
    prob, poses = gpu.problem_from_graph(g)
    prob.solver_begin(gpu.SolverOptions(max_num_iterations=100, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=2))
    prob.trace_start(4000)
    ran, done = prob.solver_step(10)
    rec, host_launches, host_seconds = prob.trace_read()
    s = prob.solver_end()
    assert ran == 10 and s.cg_form == 3
    ops = [int(o) for o in rec[:, 0]]
    while ops and ops[-1] == 0:
        ops.pop()                                                       # launches enqueued behind the pause
    HEAD, W0, CG, TAIL, LIN = 1, 2, 3, 4, 5
    i, decisions, cg_launches = 0, 0, []
    while i < len(ops):
        assert ops[i] == HEAD, (i, ops[max(0, i - 3): i + 3])
        i += 1
        if i == len(ops):
            break                                                       # the head that only finished the last accepted step
        assert ops[i] == W0
        i += 1
        n = 0
        while ops[i] == CG:
            n += 1
            i += 1
        assert n >= 2 and ops[i] == TAIL                                # at least one iteration + the launch that multiplies A x
        cg_launches.append(n - 1)
        decisions += 1
        i += 1
        if i < len(ops) and ops[i] == LIN:
            i += 1
    assert decisions == 10
    its = s.iterations
    assert cg_launches == [int(x) for x in its["linear_solver_iterations"][1:11]]        # one launch per CG iteration
    assert int((np.array(ops) == LIN).sum()) == int(its["step_is_successful"][1:11].sum())
    assert (rec[: len(ops), 2] >= rec[: len(ops), 1]).all() and (np.diff(rec[: len(ops), 1]) > 0).all()
    assert host_launches >= len(ops) and host_seconds > 0
