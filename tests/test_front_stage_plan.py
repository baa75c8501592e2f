"""CPU: the stage plan of the single-launch multifrontal factorisation (pgo_front.cpp, FrontStages) is a topological order —
every work-group of every stage a stage waits for holds an earlier ticket, which is what makes the in-kernel waits free of
deadlock whatever part of the grid is resident — and covers every work-group of the launch schedule exactly once.  Checked by the
host-only driver tools/front_check_cli (which also executes the schedule with scalar loops and checks |A x - b|)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", TOOLS, "front_check_cli"], stdout=subprocess.DEVNULL)
    return os.path.join(TOOLS, "front_check_cli")


def _edges(tmp_path, name, n, ia, ib):
    path = str(tmp_path / (name + ".txt"))
    with open(path, "w") as f:
        f.write("%d %d\n" % (n, len(ia)))
        f.write("".join("%d %d\n" % (a, b) for a, b in zip(ia, ib)))
    return path


@pytest.mark.parametrize("name", ["manhattan", "sphere", "kitti_dense", "forest"])
def test_stage_plan_is_a_topological_order(cli, ds, tmp_path, name):
    if name == "manhattan":
        g = ds.manhattan_se3(3000, 10000, seed=5)
        n, ia, ib = g.N, g.ia, g.ib
    elif name == "sphere":
        g = ds.sphere_layers(n_spheres=3, rings=16, per_ring=16)
        n, ia, ib = g.N, g.ia, g.ib
    elif name == "kitti_dense":
        kz = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
        offs = kz["cand_offsets"]
        cands = {int(key): kz["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(kz["cand_keys"])}
        g = ds.graph_from_candidates(kz["origin"], cands, seed=20260929)
        n, ia, ib = g.N, g.ia, g.ib
    else:   # three disconnected meshes: several roots
        g = ds.manhattan_se3(800, 2600, seed=9)
        n = 3 * g.N
        ia = np.concatenate([g.ia + c * g.N for c in range(3)])
        ib = np.concatenate([g.ib + c * g.N for c in range(3)])
    out = subprocess.run([cli, _edges(tmp_path, name, n, ia, ib)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "topological order ok" in out.stdout
    assert "relative residual" in out.stdout


def test_small_front_numbering_puts_children_first(cli, tmp_path):
    """The single-launch small-front factorisation takes fronts in the order of their numbers: a front's children must have
    smaller numbers (KITTI-00 replay, every front <= 96 scalars)."""
    kz = np.load(os.path.join(ROOT, "tests", "golden", "kitti00.npz"))
    n = int(max(kz["ia"].max(), kz["ib"].max())) + 1
    env = dict(os.environ, SMALL_MAX="96")
    out = subprocess.run([cli, _edges(tmp_path, "kitti", n, kz["ia"], kz["ib"])], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "children before parents ok" in out.stdout
