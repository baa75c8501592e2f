"""The `namespace ceres` facade (include/ceres/): host-only self test on CPU, and on the GPU the reference's
Build/Solve/OutputPoses flow (tools/pose_graph_solve.cpp) on the KITTI-00 replay graph against the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def tools(pkg):
    subprocess.check_call(["make", "-C", TOOLS], stdout=subprocess.DEVNULL)
    return TOOLS


def test_facade_selftest(tools):
    out = subprocess.run([os.path.join(tools, "facade_selftest")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("recover", "autodiff", "parameterization", "problem", "motion_estimate"):
        assert "OK " + name in out.stdout


@pytest.mark.gpu
def test_facade_selftest_on_the_gpu(tools, gpu):
    """The same binary with a device present: ceres::Solve succeeds on the pose-graph problem and on the MotionEstimate
    problem (REF/src/MotionEstimate.cc:71-129) built through ceres::Problem (translation recovered, rotation untouched)."""
    out = subprocess.run([os.path.join(tools, "facade_selftest")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK problem" in out.stdout and "OK motion_estimate" in out.stdout and "   t = " in out.stdout


def test_reference_flow_fails_loudly_without_gpu(tools, pkg, ds, tmp_path):
    if pkg.device_count() > 0:
        pytest.skip("GPU present")
    g = ds.manhattan_se3(30, 60, seed=1, loop_radius=5.0, min_gap=3)
    src, dst = tmp_path / "in.g2o", tmp_path / "out.txt"
    ds.write_g2o(str(src), g, exact=True)
    out = subprocess.run([os.path.join(tools, "pose_graph_solve"), str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "no CPU fallback" in out.stdout
    ids, poses = ds.read_poses(str(dst))           # OutputPoses still writes the (unchanged) poses
    assert np.allclose(poses, g.poses, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cgnr", "cholesky"])
def test_reference_flow_on_kitti00_replay(tools, gpu, ds, O, tmp_path, mode):
    k = np.load(os.path.join(G, "kitti00.npz"))
    g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
    src, dst = tmp_path / "kitti.g2o", tmp_path / "out.txt"
    ds.write_g2o(str(src), g, exact=True)
    max_it = 30 if mode == "cgnr" else 1000
    args = [os.path.join(tools, "pose_graph_solve"), str(src), str(dst), str(max_it)] + (["cgnr"] if mode == "cgnr" else [])
    out = subprocess.run(args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    import re
    assert re.search(r"Residual blocks\s+5179", out.stdout)
    ids, poses = ds.read_poses(str(dst))
    assert len(ids) == 4541 and np.allclose(poses[0], k["origin"][0], rtol=1e-5, atol=1e-12)
    og = O.Graph(k["origin"], k["ia"], k["ib"], k["meas"], None)
    if mode == "cgnr":
        op, osum, _ = O.solve(og, O.default_options(max_num_iterations=max_it, linear_solver=1))
    else:
        op, osum, _ = O.solve(og, O.default_options(max_num_iterations=max_it, linear_solver=0))
    final = [l for l in out.stdout.splitlines() if l.startswith("Final")][0]
    assert float(final.split()[-1]) == pytest.approx(osum.final_cost, rel=1e-4)
    # text output carries 6 significant digits
    assert np.abs(poses[:, :3] - op[:, :3]).max() < (2e-2 if mode == "cgnr" else 5e-2)
