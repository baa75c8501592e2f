#!/bin/bash
# End-to-end wall time of the drop-in flow (read g2o -> ceres::Problem -> ceres::Solve -> OutputPoses) on the GPU box.
# usage: tools/facade_timing.sh   (writes the KITTI-00 replay graph from tests/golden and runs tools/pose_graph_solve on it)
set -e
cd "$(dirname "$0")/.."
python - <<PY
import numpy as np, pgo_loader
ds = pgo_loader.datasets()
k = np.load("tests/golden/kitti00.npz")
g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
ds.write_g2o("/tmp/kitti00_replay.g2o", g, exact=True)
g2 = ds.manhattan_se3(10000, 40000)
ds.write_g2o("/tmp/manhattan10k.g2o", g2, exact=True)
PY
for f in kitti00_replay manhattan10k; do
  for mode in "" cgnr; do
    for rep in 1 2; do
      t0=$(date +%s.%N)
      tools/pose_graph_solve /tmp/$f.g2o /tmp/$f.out 1000 $mode > /tmp/$f.log 2>&1
      t1=$(date +%s.%N)
      echo "== $f [$mode] run $rep: wall $(python -c "print('%.3f' % ($t1 - $t0))") s"
      grep -E "Time|time|iterations|Termination|Cost|cost|Linear solver" /tmp/$f.log | head -24
    done
  done
done
