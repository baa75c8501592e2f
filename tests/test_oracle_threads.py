"""The CPU oracle's threaded mode (oracle_options.num_threads, the "all cores" leg of bench.py's cpu_baseline — SURVEY.md 8d)
must be the SAME computation as its sequential mode: Jacobian / cost evaluation split by owner, the numeric Cholesky over
independent subtrees of the elimination tree, every sum in its sequential order.  Bit-identical traces and poses."""
import numpy as np
import pytest


@pytest.mark.parametrize("name", ["manhattan", "sphere", "chain"])
def test_threaded_oracle_equals_sequential_bit_for_bit(O, ds, name):
    g = {"manhattan": lambda: ds.manhattan_se3(1500, 6000, seed=3),
         "sphere": lambda: ds.sphere_layers(n_spheres=2, rings=16, per_ring=16),
         "chain": lambda: ds.manhattan_se3(1200, 1400, seed=5)}[name]()
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    p1, s1, t1 = O.solve(og, O.default_options(max_num_iterations=12, linear_solver=0, num_threads=1))
    for nt in (2, 5, 8):
        for _ in range(2):       # repeated: a race would not repeat itself
            p, s, t = O.solve(og, O.default_options(max_num_iterations=12, linear_solver=0, num_threads=nt))
            assert np.array_equal(t, t1) and np.array_equal(p, p1), (name, nt)
    pc1, sc1, tc1 = O.solve(og, O.default_options(max_num_iterations=8, linear_solver=1, pcg_cluster=2, num_threads=1))
    pc, sc, tc = O.solve(og, O.default_options(max_num_iterations=8, linear_solver=1, pcg_cluster=2, num_threads=6))
    assert np.array_equal(tc, tc1) and np.array_equal(pc, pc1)
    assert sc.num_linear_iterations == sc1.num_linear_iterations > 0      # (the PCG's block SpMV and Jacobi blocks run on the pool as well)
