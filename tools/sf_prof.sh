# development aid: kernel times of the small-front plan (PGO_SFRONT=1) on the KITTI-00 replay; arguments: PGO_SF_DBG values
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PGO_SFRONT=1 python -m pytest tests/test_gpu_front.py -x -q -k small_front 2>&1 | tail -1
for d in ${@:-0}; do
PGO_SFRONT=1 PGO_SF_DBG=$d rocprofv3 --kernel-trace -d /tmp/sft$d -o t -- python tools/kitti_phases.py > /tmp/sflog.txt 2>&1
python tools/rocprof_summary.py /tmp/sft$d/t_results.db /tmp/sfstats.csv > /dev/null
python - $d <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/sfstats.csv')):
    if 'k_sfront' in r['kernel']: print('dbg', sys.argv[1], r['kernel'].split('(')[0].split('::')[-1], r['calls'], r['avg_us'], r['median_us'])
PY
done
