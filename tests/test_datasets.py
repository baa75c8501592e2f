"""CPU: synthetic generators and g2o text format (data formats either side of the path)."""
import numpy as np
import pytest


def test_manhattan_generator_is_seeded_and_consistent(ds, O):
    a = ds.manhattan_se3(500, 1800, seed=3)
    b = ds.manhattan_se3(500, 1800, seed=3)
    assert a.E == 1800 and a.N == 500
    assert np.array_equal(a.poses, b.poses) and np.array_equal(a.meas, b.meas) and np.array_equal(a.ia, b.ia)
    # odometry edges follow the reference direction: id_begin = current, id_end = previous (finial.cpp:211-213)
    assert np.array_equal(a.ia[:499], np.arange(1, 500)) and np.array_equal(a.ib[:499], np.arange(0, 499))
    assert (a.ia[499:] - a.ib[499:] > 20).all()
    # dead reckoning satisfies the odometry edges exactly
    og = O.Graph(a.poses, a.ia[:499], a.ib[:499], a.meas[:499], None)
    assert O.cost(og, loss_kind=0) < 1e-20
    # noise-free measurements vanish at the ground truth
    c = ds.manhattan_se3(200, 600, seed=4, sigma_t=0.0, sigma_r=0.0, identity_information=True)
    assert O.cost(O.Graph(c.truth, c.ia, c.ib, c.meas, None), loss_kind=0) < 1e-20


def test_g2o_roundtrip(ds, tmp_path):
    g = ds.manhattan_se3(60, 150, seed=2)
    p = tmp_path / "g.g2o"
    ds.write_g2o(str(p), g)
    h = ds.read_g2o(str(p))
    assert h.N == 60 and h.E == 150 and h.fixed == [0]
    assert np.array_equal(h.ia, g.ia) and np.array_equal(h.ib, g.ib)
    assert np.allclose(h.meas, g.meas, rtol=1e-5, atol=1e-9)
    assert np.allclose(h.sqrt_info, g.sqrt_info, rtol=1e-5)


def test_sphere_generator(ds):
    g = ds.sphere_layers(n_spheres=2, rings=10, per_ring=10, n_edges=900, seed=1)
    assert g.N == 200 and g.E == 900
    assert (g.ia != g.ib).all()


def test_g2o_reader_on_the_reference_file_excerpt(ds, O):
    """tests/golden/g2o_00_excerpt.g2o holds lines of the reference's own POSE_GRAPH/result/g2o/00.g2o verbatim.  Its
    edges were written from trajectory-relative transforms, so with the right reading of the format (edge i j =
    pose of j in the frame of i -> id_begin = i, id_end = j; information upper triangle; FIX) the graph is
    consistent at the file's vertex values; with the direction swapped it is not."""
    import os
    g = ds.read_g2o(os.path.join(os.path.dirname(__file__), "golden", "g2o_00_excerpt.g2o"))
    assert (g.N, g.E) == (200, 201) and g.fixed == [0] and g.sqrt_info is None
    assert np.array_equal(g.ids, np.arange(200))
    assert np.allclose(g.poses[0], [0, 0, 0, 0, 0, 0, 1])
    right = O.cost(O.Graph(g.poses, g.ia, g.ib, g.meas, None))
    swapped = O.cost(O.Graph(g.poses, g.ib, g.ia, g.meas, None))
    assert right < 1e-2 and swapped > 1e2
    # two edges of the excerpt skip a frame (the reference's accepted loop edges inside this window)
    assert sorted(zip(g.ia[np.abs(g.ia - g.ib) > 1].tolist(), g.ib[np.abs(g.ia - g.ib) > 1].tolist())) == [(145, 147), (186, 188)]


def test_generator_rejects_impossible_loop_counts(ds):
    with pytest.raises(ValueError):
        ds.manhattan_se3(30, 200)                 # 30 poses admit 45 pairs with an id gap > 20
    g = ds.manhattan_se3(30, 29)                  # odometry only
    assert (g.N, g.E) == (30, 29)


def test_pose_landmark_toy_and_its_pose_graph_form(ds, O):
    """SURVEY 8f row 3 toy: observations are consistent with the truth to the stated noise, the pose-graph statement of the problem
    (points = nodes with a constant identity quaternion, observations = between-factors without rotation information) is what the
    oracle solves, and solving it brings the points from metres to centimetres of the truth."""
    import numpy as np
    pl = ds.pose_landmark_toy(n_poses=60, n_points=400, seed=11)
    g = pl.graph
    zt = ds.qrot(ds.qconj(g.truth[pl.obs_pose, 3:]), pl.truth_points[pl.obs_point] - g.truth[pl.obs_pose, :3])
    assert np.abs(zt - pl.obs_z).max() < 6 * 0.03
    full, cmask = pl.as_pose_graph()
    assert full.N == 460 and full.E == g.E + len(pl.obs_pose) and (cmask[60:] == 2).all() and cmask[0] == 3
    assert np.all(full.sqrt_info[g.E:].reshape(-1, 6, 6)[:, 3:, :] == 0.0) and np.all(full.sqrt_info[g.E:].reshape(-1, 6, 6)[:, :, 3:] == 0.0)
    og = O.Graph(full.poses, full.ia, full.ib, full.meas, full.sqrt_info, cmask)
    p, s, tr = O.solve(og, O.default_options(max_num_iterations=40, linear_solver=0))
    assert s.termination_type == 0 and s.final_cost < 0.2 * s.initial_cost
    assert np.array_equal(p[60:, 3:], np.tile([0.0, 0.0, 0.0, 1.0], (400, 1)))          # the constant blocks stayed put
    assert np.abs(p[60:, :3] - pl.truth_points).max() < 0.25 * np.abs(pl.points - pl.truth_points).max()
