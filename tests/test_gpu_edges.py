"""-m gpu: graph construction behind the C ABI (pgo_build_odometry_edges on the GPU, pgo_build_edges' acceptance rules in
C++; finial.cpp:162-293, converter.cc:150-155, 221-234) against posegraph-ceres_amd/loop_edges.py, BIT FOR BIT, on the replay
of the reference's committed artefacts (input trajectory, Edge_Candidates_index.txt, edges_for_loop.txt in kitti00.npz)."""
import importlib
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _Twc32(pose):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_quat(pose[3:]).as_matrix()
    T[:3, 3] = pose[:3]
    return T.astype(np.float32).astype(np.float64)      # the reference keeps camera poses in CV_32F


def test_edges_for_loop_replay_bit_for_bit(gpu):
    le = importlib.import_module("posegraph_ceres_amd.loop_edges")
    k = np.load(os.path.join(GOLD, "kitti00.npz"))
    origin, loops = k["origin"], [tuple(r) for r in k["loops"]]
    accepted = set(loops)
    offs = k["cand_offsets"]
    cands = {int(key): k["cand_flat"][offs[i]:offs[i + 1]].tolist() for i, key in enumerate(k["cand_keys"])}
    rng = np.random.default_rng(3)
    rec = {pair: dict(nmatches=300, inliers=150, rvec=(rng.normal(size=3) * 0.05).tolist(), tvec=(rng.normal(size=3) * 0.1).tolist())
           for pair in loops}

    def vision(cur, prev):
        if (cur, prev) in accepted:
            return rec[(cur, prev)]
        return dict(nmatches=100, inliers=0, rvec=[0.0, 0.0, 0.0], tvec=[0.0, 0.0, 0.0])

    Twc = np.array([_Twc32(p) for p in origin])
    b = le.LoopEdgeBuilder()
    for i in range(origin.shape[0]):
        b.add_frame(i, Twc[i], cands.get(i, ()), vision)
    ia0, ib0, m0 = b.edges()
    ia, ib, m, ll = gpu.build_edges(Twc, cands, vision)
    assert len(ia) == 4540 + 639
    assert np.array_equal(ia, ia0) and np.array_equal(ib, ib0)
    assert np.array_equal(m, m0)                                            # bit for bit, odometry (GPU) and loop edges (host)
    assert [tuple(r) for r in ll.tolist()] == b.loop_list == loops          # edges_for_loop.txt, in file order
    # the batched kernel alone
    odo = gpu.build_odometry_edges(Twc)
    sel = (ia - ib) == 1
    assert np.array_equal(odo[ib[sel]], m[sel])
    assert np.allclose(np.linalg.norm(odo[:, 3:], axis=1), 1.0, atol=1e-6)


def test_acceptance_rules_match_python(gpu):
    le = importlib.import_module("posegraph_ceres_amd.loop_edges")
    T = np.tile(np.eye(4), (9, 1, 1))
    good = dict(nmatches=281, inliers=101, rvec=[0.0, 0.1, 0.0], tvec=[0.2, 0.0, 0.0])
    table = {(6, 0): dict(good, nmatches=280), (6, 1): dict(good, inliers=100), (6, 2): dict(good, tvec=[0.7, 0.0, 0.0]),
             (6, 3): good, (6, 4): good, (7, 6): good, (7, 3): good, (8, 6): good, (8, 2): good}
    cands = {6: [5, 0, 1, 2, 3, 4], 7: [6, 3], 8: [7, 6, 2]}
    b = le.LoopEdgeBuilder()
    for i in range(9):
        b.add_frame(i, T[i], cands.get(i, ()), lambda c, p: table.get((c, p)))
    ia0, ib0, m0 = b.edges()
    ia, ib, m, ll = gpu.build_edges(T, cands, lambda c, p: table.get((c, p)))
    assert list(zip(ia.tolist(), ib.tolist())) == [(6, 5), (6, 3), (7, 6), (7, 3), (8, 7), (8, 2)]
    assert np.array_equal(ia, ia0) and np.array_equal(ib, ib0) and np.array_equal(m, m0) and len(ll) == 0
    # loop list gap
    ia, ib, m, ll = gpu.build_edges(T, cands, lambda c, p: table.get((c, p)), gpu.EdgeRules(loop_list_gap=2))
    assert [tuple(r) for r in ll.tolist()] == [(6, 3), (7, 3), (8, 2)]
