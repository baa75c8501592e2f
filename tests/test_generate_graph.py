"""CPU test of tools/generate_graph.cpp — the C++ statement of the synthetic generators of SURVEY.md §8d (the Python generators
of datasets.py stay the source of the committed fixtures; the two draw from different random streams).  The files it writes
must be well-formed g2o graphs of the requested shape whose measurements are consistent with ONE trajectory: the CPU oracle
brings the dead-reckoning start down to a cost of ~1.5-3.5 per edge (6 noisy components per edge at unit information
weight, Huber, minus the poses' degrees of freedom)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "generate_graph")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "generate_graph"])
    return TOOL


@pytest.mark.parametrize("args,n,e,gap", [(["manhattan", "400", "900", "11"], 400, 900, 20),
                                           (["sphere", "2", "10", "12", "900", "5"], 240, 900, 12)])
def test_generated_graph_is_consistent(tool, ds, O, tmp_path, args, n, e, gap):
    out = str(tmp_path / "g.g2o")
    subprocess.check_call([tool] + args + [out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    g = ds.read_g2o(out)
    ia, ib = np.asarray(g.ia), np.asarray(g.ib)
    assert g.N == n and len(ia) == e
    assert np.array_equal(ia[:n - 1], np.arange(1, n)) and np.array_equal(ib[:n - 1], np.arange(0, n - 1))   # the odometry chain first
    rest = np.stack([ia[n - 1:], ib[n - 1:]], 1)
    assert np.all(rest[:, 0] > rest[:, 1]) and len({(a, b) for a, b in rest}) == len(rest)                     # id_begin > id_end, no duplicates
    if args[0] == "manhattan":
        assert np.all(rest[:, 0] - rest[:, 1] > gap)
    assert np.allclose(np.linalg.norm(g.poses[:, 3:], axis=1), 1.0) and np.allclose(np.linalg.norm(g.meas[:, 3:], axis=1), 1.0)
    L = g.sqrt_info.reshape(-1, 6, 6)
    assert np.allclose(L[:, range(6), range(6)], [20.0] * 3 + [100.0] * 3) and np.count_nonzero(L[0]) == 6   # chol(diag(1/sigma^2))
    og = O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info)
    _, s, _ = O.solve(og, O.default_options(max_num_iterations=100, linear_solver=0))
    assert s.termination_type == 0 and s.final_cost < 0.2 * s.initial_cost
    assert 1.2 < s.final_cost / e < 3.5


def test_same_seed_same_file(tool, tmp_path):
    a, b = str(tmp_path / "a.g2o"), str(tmp_path / "b.g2o")
    for out in (a, b):
        subprocess.check_call([tool, "manhattan", "300", "600", "3", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(a).read() == open(b).read()
