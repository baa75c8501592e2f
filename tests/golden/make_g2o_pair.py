"""Stores the one replayable before/after pair the reference holds for an SE(3) pose-graph solve as arrays
(DATA only).  Run in the BUILD container (reads /root/reference):

    python tests/golden/make_g2o_pair.py

REF = /root/reference/src/POSE_GRAPH
  * result/g2o/result_before.g2o  — graph as handed to the optimiser: 4541 VERTEX_SE3:QUAT, FIX 0, 4695 EDGE_SE3:QUAT
                                    (4540 odometry + 155 loop edges, identity information), written by
                                    test/pose_graph_try1.cpp:137
  * result/g2o/result_after.g2o   — the same graph after `optimizer.optimize(1000)` (g2o Levenberg-Marquardt,
                                    RobustKernelHuber on every edge), written by test/pose_graph_try1.cpp:147-148
    (edges of the two files are identical; only the vertices moved: mean 3.3 mm, max 8.7 mm)
  * result/g2o/111                — a reference-held INPUT graph with real loop-closure measurements and large
                                    residuals (2761 vertices, 8900 edges; Huber active on most loop edges); there is
                                    no "after" for it, it is used as an input for GPU-vs-oracle parity only.

  * result/result_before.g2o -> result/result_after.g2o (ONE DIRECTORY UP; same vertices and edge ids as the pair
                                    above, different edge measurements): the far sharper pair.  At *before* the cost
                                    is 5359 with 4056 of the 4695 edges in Huber's linear region; *after* lies 480 m
                                    (mean) / 931 m (max) away at cost 2.99939 with every edge carrying residual.
                                    Stored as g2o_pair_strong.npz; tests/test_g2o_strong_pair.py states what it pins.

Both files print 6 significant digits.  g2o's EdgeSE3 error is toVectorMQT(Z^-1 Xi^-1 Xj) = [t ; vec q] of the
error transform, so with identity information chi2 = |dt|^2 + |vec dq|^2; this repo's residual
(PoseGraph3dError.h:32-51) has the rotation part 2 vec(dq), hence sqrt-information L = diag(1,1,1,.5,.5,.5)
makes the two costs the same function of the poses (tests/test_g2o_pair.py states what follows from that).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

ds = pgo_loader.datasets()
REF = "/root/reference/src/POSE_GRAPH/result/g2o"


def main():
    gb = ds.read_g2o(os.path.join(REF, "result_before.g2o"))
    ga = ds.read_g2o(os.path.join(REF, "result_after.g2o"))
    assert gb.N == ga.N == 4541 and gb.E == ga.E == 4695 and gb.fixed == ga.fixed == [0]
    assert np.array_equal(gb.ia, ga.ia) and np.array_equal(gb.ib, ga.ib) and np.array_equal(gb.meas, ga.meas)
    assert gb.sqrt_info is None and ga.sqrt_info is None      # identity information on every edge
    assert np.array_equal(gb.ids, np.arange(gb.N))
    np.savez_compressed(os.path.join(HERE, "g2o_pair.npz"), before=gb.poses, after=ga.poses, ia=gb.ia, ib=gb.ib,
                        meas=gb.meas, fixed=np.array(gb.fixed, dtype=np.int32))
    d = np.linalg.norm(ga.poses[:, :3] - gb.poses[:, :3], axis=1)
    print("g2o pair: %d vertices, %d edges (%d loop), moved mean %.2f mm max %.2f mm" % (
        gb.N, gb.E, int((np.abs(gb.ia - gb.ib) > 1).sum()), 1e3 * d.mean(), 1e3 * d.max()))

    up = os.path.dirname(REF)
    sb = ds.read_g2o(os.path.join(up, "result_before.g2o"))
    sa = ds.read_g2o(os.path.join(up, "result_after.g2o"))
    assert sb.N == sa.N == 4541 and sb.E == sa.E == 4695 and sb.fixed == sa.fixed == [0]
    assert np.array_equal(sb.ia, sa.ia) and np.array_equal(sb.ib, sa.ib) and np.array_equal(sb.meas, sa.meas)
    assert np.array_equal(sb.ia, gb.ia) and np.array_equal(sb.ib, gb.ib) and not np.array_equal(sb.meas, gb.meas)
    assert sb.sqrt_info is None and sa.sqrt_info is None
    np.savez_compressed(os.path.join(HERE, "g2o_pair_strong.npz"), before=sb.poses, after=sa.poses, ia=sb.ia, ib=sb.ib,
                        meas=sb.meas, fixed=np.array(sb.fixed, dtype=np.int32))
    d = np.linalg.norm(sa.poses[:, :3] - sb.poses[:, :3], axis=1)
    print("strong pair: %d vertices, %d edges, moved mean %.1f m max %.1f m" % (sb.N, sb.E, d.mean(), d.max()))

    g = ds.read_g2o(os.path.join(REF, "111"))
    assert g.sqrt_info is None and g.fixed == [0]
    np.savez_compressed(os.path.join(HERE, "g2o_111.npz"), poses=g.poses, ia=g.ia, ib=g.ib, meas=g.meas,
                        ids=g.ids.astype(np.int32), fixed=np.array(g.fixed, dtype=np.int32))
    print("g2o 111: %d vertices, %d edges" % (g.N, g.E))


if __name__ == "__main__":
    main()
