#!/bin/bash
# Development aid (r06): A/B of two builds of the library on one box — posegraph-ceres_amd/libpgo_hip.<a>.so against libpgo_hip.<b>.so:
# headline bench (resident stream), the fused / two-kernel streams (tools/fused_time.py), C4 on one GPU (tools/c4_lm.py).
A=${1:-licm_off}; B=${2:-licm_on}
P=posegraph-ceres_amd
for round in 1 2; do
for v in $A $B; do
  cp $P/libpgo_hip.$v.so $P/libpgo_hip.so; touch $P/libpgo_hip.so
  python bench.py --steps 20 --warmup 5 --no-c4-kernels --no-cpu-baseline --no-quality --no-exact-blocks 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'ms_per_step', d['ms_per_step'], 'us/turn', d.get('us_per_cg_turn'), 'fixed', d.get('fixed_us_per_lm_iteration'))
"
done
done
for v in $A $B $A $B; do
  echo "== $v fused / two-kernel"
  PGO_AB_LIB=$PWD/$P/libpgo_hip.$v.so python tools/fused_time.py 256 2>&1 | tail -2
done
for v in $A $B; do
  cp $P/libpgo_hip.$v.so $P/libpgo_hip.so; touch $P/libpgo_hip.so
  echo "== $v C4"
  python tools/c4_lm.py 2>&1 | tail -6
done
cp $P/libpgo_hip.$A.so $P/libpgo_hip.so
