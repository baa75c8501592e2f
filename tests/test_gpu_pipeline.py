"""-m gpu tests of the device-resident Levenberg-Marquardt loop (r03): the trust-region decisions of SURVEY.md A.6 steps 4-7
are taken by the last work-group of the step tail and the host enqueues the kernel sequences of the next iterations ahead of
them (pgo_kernels.h LmDev, pgo_lm_rules.h, pgo_solver.cpp lm_run_pipelined).  The host-in-the-loop driver of r02 is still
there (several ranks, batched solve, PGO_NO_PIPELINE=1) and applies the very same rule function, so the two must produce the
same iteration records BIT FOR BIT: same kernels, same inputs, same decisions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["iteration", "step_is_successful", "linear_solver_iterations", "cost", "cost_change", "gradient_max_norm",
          "step_norm", "relative_decrease", "trust_region_radius"]


def _solve(gpu, g, host_loop, **opt):
    old = os.environ.get("PGO_NO_PIPELINE")
    os.environ["PGO_NO_PIPELINE"] = "1" if host_loop else "0"
    os.environ["PGO_PIPELINE_PCG"] = "1"      # (PCG keeps the host in the loop by default: see pipeline_wanted in pgo_solver.cpp)
    try:
        prob, poses = gpu.problem_from_graph(g)
        s = gpu.solve(gpu.SolverOptions(**opt), prob)
    finally:
        os.environ.pop("PGO_PIPELINE_PCG", None)
        if old is None:
            os.environ.pop("PGO_NO_PIPELINE", None)
        else:
            os.environ["PGO_NO_PIPELINE"] = old
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
    a, pa = _solve(gpu, g, False, **opt)
    b, pb = _solve(gpu, g, True, **opt)
    assert len(a.iterations) > 5
    _same(a, b, pa, pb)


@pytest.mark.parametrize("limit", [1, 2, 7])
def test_iteration_limit_and_tolerances(gpu, ds, limit):
    """max_num_iterations ends both drivers at the same record with NO_CONVERGENCE; so do loose tolerances with their reasons."""
    g = ds.manhattan_se3(600, 2000, seed=5)
    a, pa = _solve(gpu, g, False, max_num_iterations=limit, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    b, pb = _solve(gpu, g, True, max_num_iterations=limit, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    assert a.termination_type == gpu.NO_CONVERGENCE and len(a.iterations) == limit + 1
    _same(a, b, pa, pb)
    for extra in (dict(function_tolerance=1e-2), dict(parameter_tolerance=1e-3), dict(gradient_tolerance=1e-1),
                  dict(min_trust_region_radius=1e5, initial_trust_region_radius=1e5)):
        a, pa = _solve(gpu, g, False, max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, **extra)
        b, pb = _solve(gpu, g, True, max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY, **extra)
        assert a.termination_type == gpu.CONVERGENCE, extra
        _same(a, b, pa, pb)


def test_stepping_in_pieces_equals_one_solve(gpu, ds):
    """pgo_solver_begin / step(n) / end in uneven pieces (what bench.py times) gives the records of one pgo_solve; a reset
    restarts the same trajectory."""
    g = ds.manhattan_se3(800, 2800, seed=9)
    opt = gpu.SolverOptions(max_num_iterations=40, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    prob, poses = gpu.problem_from_graph(g)
    whole = gpu.solve(opt, prob)
    os.environ["PGO_PIPELINE_PCG"] = "1"
    try:
        prob2, poses2 = gpu.problem_from_graph(g)
        prob2.solver_begin(opt)
        total = 0
        for n in (1, 3, 2, 5, 40):
            ran, done = prob2.solver_step(n)
            total += ran
            if done:
                break
        s = prob2.solver_end()
        prob3, poses3 = gpu.problem_from_graph(g)
        prob3.solver_begin(opt)
        prob3.solver_step(6)
        prob3.solver_reset()
        ran3, done3 = prob3.solver_step(6)
        s3 = prob3.solver_end()
    finally:
        os.environ.pop("PGO_PIPELINE_PCG", None)
    assert len(s.iterations) == len(whole.iterations)
    for f in FIELDS:
        assert np.array_equal(s.iterations[f], whole.iterations[f]), f
    assert np.array_equal(poses, poses2)
    # reset: the first 6 iterations again, identical to the first 6 of the solve
    assert ran3 == 6 and not done3
    last6 = s3.iterations[-6:]
    for f in ("cost", "step_is_successful", "linear_solver_iterations", "trust_region_radius"):
        assert np.array_equal(last6[f], whole.iterations[f][1:7]), f
