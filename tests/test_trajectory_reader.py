"""pgo_read_trajectory (C ABI, host only): GroundTruth::loadPoses1 / loadPoses2 (REF/src/GroundTruth.cc:22-73) including the
reference's quaternion scramble, checked against what the reference's own committed output shows:
trajectory_origin.txt row 0 prints q = (1, 0, ~0, ~0) for an input pose whose file quaternion is the identity."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_format1_scramble_matches_reference_row0(pkg, tmp_path):
    import importlib
    le = importlib.import_module("posegraph_ceres_amd.loop_edges")
    f = tmp_path / "traj1.txt"
    f.write_text("0 0 0 0 0 0 1\n1.5 -2 3 0 0 0.70710678118654757 0.70710678118654757\n")
    T = pkg.read_trajectory(f, 1)
    assert T.shape == (2, 4, 4)
    # identity file quaternion -> Eigen (x, y, z, w) = (1, 0, 0, 0): a half turn about x, R = diag(1, -1, -1)
    assert np.array_equal(T[0], np.diag([1.0, -1.0, -1.0, 1.0]))
    # ... which Converter::toPose3d prints as the first row of the reference's trajectory_origin.txt
    k = np.load(os.path.join(GOLD, "kitti00.npz"))
    q0 = le.to_pose3d(T[0])[3:]
    assert np.allclose(q0, k["origin"][0, 3:], atol=1e-6) and q0[0] == 1.0
    # second row: file (qx,qy,qz,qw) = (0,0,s,s) -> Eigen (x,y,z,w) = (s,0,0,s): a quarter turn about x (not about z)
    s = np.float32
    assert np.array_equal(T[1][:3, 3], np.array([1.5, -2.0, 3.0]))
    R = T[1][:3, :3]
    assert np.allclose(R, [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-7)
    assert np.array_equal(R, R.astype(s).astype(np.float64))          # float32 values, as the reference's CV_32F matrices


def test_format2_kitti_rows(pkg, tmp_path):
    rng = np.random.default_rng(0)
    P = rng.normal(size=(5, 12))
    f = tmp_path / "traj2.txt"
    f.write_text("\n".join(" ".join("%.17g" % v for v in row) for row in P) + "\n")
    T = pkg.read_trajectory(f, 2)
    assert T.shape == (5, 4, 4)
    assert np.array_equal(T[:, :3, :].reshape(5, 12), P.astype(np.float32).astype(np.float64))
    assert np.array_equal(T[:, 3, :], np.tile([0.0, 0.0, 0.0, 1.0], (5, 1)))
    # an incomplete last row is not a pose; a missing file is an error with a message
    f.write_text(f.read_text() + "1 2 3\n")
    assert pkg.read_trajectory(f, 2).shape[0] == 5
    with pytest.raises(pkg.PgoError):
        pkg.read_trajectory(tmp_path / "missing.txt", 1)
    with pytest.raises(pkg.PgoError):
        pkg.read_trajectory(f, 3)
