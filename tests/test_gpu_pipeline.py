"""-m gpu tests of the device-resident Levenberg-Marquardt loop (r03): the trust-region decisions of SURVEY.md A.6 steps 4-7
are taken by the last work-group of the step tail and the host enqueues the kernel sequences of the next iterations ahead of
them (pgo_kernels.h LmDev, pgo_lm_rules.h, pgo_lm.cpp lm_run_pipelined / lm_run_universal).  The host-in-the-loop driver of r02 is still
there (several ranks, batched solve, PGO_NO_PIPELINE=1) and applies the very same rule function, so the two must produce the
same iteration records BIT FOR BIT: same kernels, same inputs, same decisions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["iteration", "step_is_successful", "linear_solver_iterations", "cost", "cost_change", "gradient_max_norm",
          "step_norm", "relative_decrease", "trust_region_radius"]


DRIVERS = {"host": {"PGO_NO_PIPELINE": "1"},                                   # r02: the host decides, one hand-off per iteration
           "seq": {"PGO_NO_PIPELINE": "0", "PGO_UNI": "0"},   # allotted sequences (default for exact steps; for PCG: + the knob pipeline_pcg, below)
           "uni": {"PGO_NO_PIPELINE": "0", "PGO_UNI": "1"}}                        # universal stream (default for PCG)


class _Env:
    def __init__(self, driver):
        self.new = DRIVERS[driver]
        self.seq = driver == "seq"

    def __enter__(self):
        import pgo_loader
        self.old = {k: os.environ.get(k) for k in ("PGO_NO_PIPELINE", "PGO_UNI")}
        for k in self.old:
            os.environ.pop(k, None)
        os.environ.update(self.new)
        pgo_loader.load().tuning_set("pipeline_pcg", 1 if self.seq else None)

    def __exit__(self, *a):
        import pgo_loader
        pgo_loader.load().tuning_set("pipeline_pcg", None)
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _solve(gpu, g, driver, **opt):
    opt.setdefault("pcg_form", 1)      # the three drivers run the same kernels only with Ceres' CG recurrences (the fused stream's own tests: test_gpu_fused.py)
    with _Env(driver):
        prob, poses = gpu.problem_from_graph(g)
        s = gpu.solve(gpu.SolverOptions(**opt), prob)
    return s, poses


def _same(a, b, pa, pb):
    assert len(a.iterations) == len(b.iterations)
    for f in FIELDS:
        assert np.array_equal(a.iterations[f], b.iterations[f]), f
    assert a.termination_type == b.termination_type and a.reason == b.reason and a.message == b.message
    assert a.final_cost == b.final_cost and a.initial_cost == b.initial_cost
    assert a.num_successful_steps == b.num_successful_steps and a.num_unsuccessful_steps == b.num_unsuccessful_steps
    assert a.num_linear_solver_iterations == b.num_linear_solver_iterations
    assert a.final_trust_region_radius == b.final_trust_region_radius
    assert a.final_gradient_max_norm == b.final_gradient_max_norm
    assert np.array_equal(pa, pb)


@pytest.mark.parametrize("name,exact,cluster", [
    ("manhattan1000", False, 2), ("manhattan1000", False, 1), ("manhattan1000", True, 1),
    ("sphere2x20", True, 1), ("sphere2x20", False, 2), ("chain", True, 1), ("chain", False, 4)])
def test_device_decisions_equal_host_decisions_bit_for_bit(gpu, ds, name, exact, cluster):
    """Whole solves to their own stop, both drivers: PCG with 6x6 / 12x12 / 24x24 Jacobi blocks (CG runs from 3 to > 100
    iterations, so sequences whose CG outlives them — continuation in the next sequence, host-enqueued continuation — occur),
    exact steps through the multifrontal and the small-front factorisations."""
    if name == "chain":
        g = ds.manhattan_se3(1500, 1700, seed=23)      # chain-like (E < 1.5 N): small fronts
    elif name == "sphere2x20":
        g = ds.sphere_layers(n_spheres=2, rings=20, per_ring=20)
    else:
        g = ds.manhattan_se3(1000, 3500, seed=17)
    ls = gpu.SPARSE_NORMAL_CHOLESKY if exact else gpu.BLOCK_JACOBI_PCG
    opt = dict(max_num_iterations=120, linear_solver_type=ls, pcg_cluster_poses=cluster)
    b, pb = _solve(gpu, g, "host", **opt)
    assert len(b.iterations) > 5
    a, pa = _solve(gpu, g, "seq", **opt)
    _same(a, b, pa, pb)
    if not exact:
        # The universal stream runs the residual refresh of every 10th CG iteration as  x += alpha p | A x | r = b - A x  (three
        # launches of its two kernels); the other two drivers multiply A (x + alpha p) with alpha re-summed in the SpMV's own
        # work-group shape, so their A x is the product with an x that differs from the stored one in the last bit.  With the
        # refresh off the three drivers are bit-identical; with it on the stream agrees to rounding over the first iterations (same CG counts,
        # same decisions) and ends at the same cost to 1e-5.
        a, pa = _solve(gpu, g, "uni", **opt)
        n = min(len(a.iterations), len(b.iterations), 9)          # (further on the last-bit difference is amplified along the LM path)
        assert np.array_equal(a.iterations["step_is_successful"][:n], b.iterations["step_is_successful"][:n])
        assert np.array_equal(a.iterations["linear_solver_iterations"][:n], b.iterations["linear_solver_iterations"][:n])
        assert np.allclose(a.iterations["cost"][:n], b.iterations["cost"][:n], rtol=1e-6, atol=0)   # (sphere: CG runs of 100+ iterations carry the bit to 3e-8)
        if a.termination_type == b.termination_type == gpu.CONVERGENCE:
            assert abs(a.final_cost - b.final_cost) <= 1e-5 * b.final_cost
        else:
            # the sphere graph is still descending around iteration 120: measured r05, the host-driven loop meets the function
            # tolerance at iteration 118 (2.4194e4), the stream is at 2.4435e4 when its 120 iterations are used up
            assert gpu.NO_CONVERGENCE in (a.termination_type, b.termination_type)
            assert abs(a.final_cost - b.final_cost) <= 2e-2 * b.final_cost
        b0, pb0 = _solve(gpu, g, "host", cg_residual_reset_period=0, **opt)
        a0, pa0 = _solve(gpu, g, "uni", cg_residual_reset_period=0, **opt)
        _same(a0, b0, pa0, pb0)


@pytest.mark.parametrize("limit", [1, 2, 7])
def test_iteration_limit_and_tolerances(gpu, ds, limit):
    """max_num_iterations ends both drivers at the same record with NO_CONVERGENCE; so do loose tolerances with their reasons."""
    g = ds.manhattan_se3(600, 2000, seed=5)
    kw = dict(max_num_iterations=limit, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, cg_residual_reset_period=0)
    b, pb = _solve(gpu, g, "host", **kw)
    for driver in ("seq", "uni"):
        a, pa = _solve(gpu, g, driver, **kw)
        assert a.termination_type == gpu.NO_CONVERGENCE and len(a.iterations) == limit + 1
        _same(a, b, pa, pb)
    for extra in (dict(function_tolerance=1e-2), dict(parameter_tolerance=1e-3), dict(gradient_tolerance=1e-1),
                  dict(min_trust_region_radius=1e5, initial_trust_region_radius=1e5)):
        a, pa = _solve(gpu, g, "seq", max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, **extra)
        b, pb = _solve(gpu, g, "host", max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, **extra)
        assert a.termination_type == gpu.CONVERGENCE, extra
        _same(a, b, pa, pb)
        kw = dict(max_num_iterations=60, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, cg_residual_reset_period=0)
        a, pa = _solve(gpu, g, "uni", **kw, **extra)
        b, pb = _solve(gpu, g, "host", **kw, **extra)
        _same(a, b, pa, pb)        # (truncated PCG may or may not get there within 60 iterations: whatever the host-driven run does)


@pytest.mark.parametrize("driver", ["seq", "uni"])
def test_stepping_in_pieces_equals_one_solve(gpu, ds, driver):
    """pgo_solver_begin / step(n) / end in uneven pieces (what bench.py times) gives the records of one pgo_solve; a reset
    restarts the same trajectory."""
    g = ds.manhattan_se3(800, 2800, seed=9)
    kw = dict(max_num_iterations=40, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, cg_residual_reset_period=0)
    opt = gpu.SolverOptions(**kw)
    whole, poses = _solve(gpu, g, "host", **kw)
    with _Env(driver):
        prob2, poses2 = gpu.problem_from_graph(g)
        prob2.solver_begin(opt)
        for n in (1, 3, 2, 5, 40):
            ran, done = prob2.solver_step(n)
            assert ran == n or done
            if done:
                break
        s = prob2.solver_end()
        prob3, poses3 = gpu.problem_from_graph(g)
        prob3.solver_begin(opt)
        prob3.solver_step(6)
        prob3.solver_reset()
        ran3, done3 = prob3.solver_step(6)
        s3 = prob3.solver_end()
    assert len(s.iterations) == len(whole.iterations)
    for f in FIELDS:
        assert np.array_equal(s.iterations[f], whole.iterations[f]), f
    assert np.array_equal(poses, poses2)
    # reset: the first 6 iterations again, identical to the first 6 of the solve
    assert ran3 == 6 and not done3
    last6 = s3.iterations[-6:]
    for f in ("cost", "step_is_successful", "linear_solver_iterations", "trust_region_radius"):
        assert np.array_equal(last6[f], whole.iterations[f][1:7]), f


@pytest.mark.parametrize("period", [5, 3, 7])
def test_allotted_sequences_with_an_odd_refresh_period(gpu, ds, period):
    """a continuation sequence is enqueued as CG iterations 1, 2, ... and picks the ping-pong buffer of p
    from that launch-time parity, so it may only take over after an EVEN number of completed iterations.  With an odd
    cg_residual_reset_period the host-enqueued continuation used to end on an odd multiple of the period (period 5: 8 -> 35)
    and the sequence behind it multiplied the wrong p.  Chain-like graph, 6x6 blocks: the early CG runs take > 40 iterations.
    The allotted-sequence driver must reproduce the host-in-the-loop driver bit for bit."""
    g = ds.manhattan_se3(1500, 1700, seed=23)
    opt = dict(max_num_iterations=25, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=1,
               cg_residual_reset_period=period)
    b, pb = _solve(gpu, g, "host", **opt)
    assert b.iterations["linear_solver_iterations"].max() > 40
    a, pa = _solve(gpu, g, "seq", **opt)
    _same(a, b, pa, pb)
