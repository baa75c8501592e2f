"""CPU: synthetic generators and g2o text format (data formats either side of the path)."""
import numpy as np


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
