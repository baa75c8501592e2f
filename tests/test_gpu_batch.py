"""-m gpu tests of pgo_solve_batch: several independent pose graphs solved as the components of one block-diagonal problem
(one launch sequence), every Levenberg-Marquardt decision per problem.  The checker is the single-problem path through the
same C ABI (itself held to the oracle by test_gpu_parity / test_gpu_golden): per problem the same number of iterations, the
same accept / reject sequence, costs to 1e-7 relative, poses to 1e-6."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _graphs(ds):
    k = np.load(os.path.join(G, "kitti00.npz"))
    return [ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None),          # identity information (the reference's)
            ds.manhattan_se3(400, 1400, seed=7),                                          # diagonal information
            ds.sphere_layers(n_spheres=2, rings=12, per_ring=12),
            ds.manhattan_se3(300, 700, seed=11),
            ds.manhattan_se3(1500, 2100, seed=5)]


def _compare(single, batch, poses_s, poses_b):
    assert len(single.iterations) == len(batch.iterations)
    assert list(single.iterations["step_is_successful"]) == list(batch.iterations["step_is_successful"])
    assert np.allclose(single.iterations["cost"], batch.iterations["cost"], rtol=1e-7)
    assert np.allclose(single.iterations["trust_region_radius"], batch.iterations["trust_region_radius"], rtol=1e-5)
    assert single.termination_type == batch.termination_type and single.c.reason == batch.c.reason
    assert batch.final_cost == pytest.approx(single.final_cost, rel=1e-7)
    assert np.abs(poses_s - poses_b).max() <= 1e-6
    assert batch.c.num_poses == single.c.num_poses and batch.c.num_edges == single.c.num_edges
    assert batch.linear_solver_used == 0 and batch.num_factorizations >= len(batch.iterations) - 1


def test_batch_follows_the_single_problem_traces(gpu, ds):
    """Five graphs of different size, topology and information kind, reference options (exact steps): every component of the
    batch reproduces its own single-problem solve, although they need different numbers of iterations."""
    gs = _graphs(ds)
    opt = gpu.SolverOptions(max_num_iterations=60, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    singles = []
    for g in gs:
        prob, poses = gpu.problem_from_graph(g)
        singles.append((gpu.solve(opt, prob), poses))
    pairs = [gpu.problem_from_graph(g) for g in gs]
    sums = gpu.solve_batch(opt, [p for p, _ in pairs])
    assert len({len(s.iterations) for s in sums}) > 1          # the components really stop at different iterations
    for (ss, ps), sb, (_, pb) in zip(singles, sums, pairs):
        _compare(ss, sb, ps, pb)


def test_batch_of_identical_graphs_and_iteration_cap(gpu, ds):
    """Eight copies of one graph give the same result eight times — to rounding: the linearisation sums a row's incidences in lane
    pairs, so where the pairs fall depends on the parity of the row's first slot inside the union — ; max_num_iterations is
    honoured per problem."""
    g = ds.manhattan_se3(500, 1200, seed=3)
    opt = gpu.SolverOptions(max_num_iterations=4, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    prob, poses = gpu.problem_from_graph(g)
    single = gpu.solve(opt, prob)
    pairs = [gpu.problem_from_graph(g) for _ in range(8)]
    sums = gpu.solve_batch(opt, [p for p, _ in pairs])
    for sb, (_, pb) in zip(sums, pairs):
        _compare(single, sb, poses, pb)
        assert np.abs(pb - pairs[0][1]).max() <= 1e-9
        assert sb.termination_type == gpu.NO_CONVERGENCE and len(sb.iterations) == 5


def test_converged_component_stops_while_the_others_go_on(gpu, ds):
    """A problem already at its optimum terminates in its first iterations; the fresh one beside it is not disturbed."""
    ga, gb = ds.manhattan_se3(400, 1400, seed=7), ds.manhattan_se3(600, 1500, seed=9)
    opt = gpu.SolverOptions(max_num_iterations=50, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    pa, poses_a = gpu.problem_from_graph(ga)
    gpu.solve(opt, pa)                                   # poses_a now at the optimum; pa keeps pointing at them
    done_alone = gpu.solve(opt, pa)                      # what a second solve from there does on its own
    at_optimum = poses_a.copy()
    pb, poses_b = gpu.problem_from_graph(gb)
    single_b = gpu.solve(opt, gpu.problem_from_graph(gb)[0])
    sa, sb = gpu.solve_batch(opt, [pa, pb])
    assert len(sa.iterations) == len(done_alone.iterations) <= 3 and len(sb.iterations) == len(single_b.iterations) > 5
    assert sa.final_cost == pytest.approx(done_alone.final_cost, rel=1e-9)
    assert np.abs(poses_a - at_optimum).max() <= 1e-6
    assert sb.final_cost == pytest.approx(single_b.final_cost, rel=1e-7)


def test_batch_refuses_what_it_does_not_serve(gpu, ds):
    g = ds.manhattan_se3(200, 500, seed=2)
    pa, _ = gpu.problem_from_graph(g)
    pb, _ = gpu.problem_from_graph(g, loss=gpu.CAUCHY)
    with pytest.raises(gpu.PgoError):
        gpu.solve_batch(gpu.SolverOptions(linear_solver_type=gpu.BLOCK_JACOBI_PCG), [pa])
    with pytest.raises(gpu.PgoError):
        gpu.solve_batch(gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY), [pa, pb])


def test_small_front_plan_with_many_workgroups_in_flight_is_reproducible(gpu, ds, monkeypatch):
    """Regression (r02): 48 KITTI-00 graphs through the small-front plan = ~14 000 LDS fronts per level, several workgroups per
    CU, so the waves of one workgroup drift apart.  A pose step of k_sfront_factor used to write the factorised pivot block
    before every wave had read the unfactorised one: a few components of such a batch came out with 13-15 iterations and a cost
    off in the 7th digit (and a single solve did so once in ~1000 runs).  The components must agree (to 1e-12; bit for bit from run to run) and follow the single solve."""
    k = np.load(os.path.join(G, "kitti00.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    opt = gpu.SolverOptions(max_num_iterations=1000, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    monkeypatch.setenv("PGO_SFRONT", "1")
    prob, _ = gpu.problem_from_graph(g)
    single = gpu.solve(opt, prob)
    assert single.c.factor_kind == 3
    runs = []
    for _ in range(2):
        pairs = [gpu.problem_from_graph(g) for _ in range(48)]
        sums = gpu.solve_batch(opt, [p for p, _ in pairs])
        assert all(s.c.factor_kind == 3 for s in sums)
        assert {len(s.iterations) for s in sums} == {len(single.iterations)}
        # among the components: to the rounding of the linearisation's pair sums (a row's pairs fall by the parity of its first slot in
        # the union: 1e-15 measured; the race moved the 7th digit)
        costs = [s.final_cost for s in sums]
        assert max(costs) - min(costs) <= 1e-12 * single.final_cost
        assert sums[0].final_cost == pytest.approx(single.final_cost, rel=1e-12)
        runs.append(costs)
    assert runs[0] == runs[1]                                               # and bit for bit from run to run
