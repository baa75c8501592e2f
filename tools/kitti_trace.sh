# development aid: kernel trace of ONE KITTI-00 exact solve (the last of six), per-kernel table + the GPU-busy share of an LM iteration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/kt -o t -- python tools/kitti_phases.py > /tmp/ktlog.txt 2>&1
tail -8 /tmp/ktlog.txt
python - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('/tmp/kt/**/t_results.db', recursive=True)[0] if glob.glob('/tmp/kt/**/t_results.db', recursive=True) else '/tmp/kt/t_results.db')
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# last solve: find the last k_sfront_upos (once per topology) and take everything after it
idx = max(i for i, r in enumerate(rows) if 'k_sfront_upos' in r[0])
rows = rows[idx:]
t0, t1 = rows[0][1], rows[-1][2]
busy = sum(e - s for _, s, e in rows)
print("last solve: %d launches, span %.3f ms, busy %.3f ms (%.0f %%)" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
agg = collections.OrderedDict()
for n, s, e in rows:
    n = n.split('(')[0].split('::')[-1]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%-34s %5d launches %9.1f us total %7.2f us avg" % (n, c, t, t / c))
# gaps
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0].split('(')[0].split('::')[-1], rows[i + 1][0].split('(')[0].split('::')[-1]) for i in range(len(rows) - 1))
tot_gap = sum(g for g, _, _ in gaps)
print("gaps between launches: total %.1f us, median %.2f us" % (tot_gap, gaps[len(gaps) // 2][0]))
by = collections.Counter()
for g_, a, b in gaps: by[(a, b)] += g_
for (a, b), g_ in by.most_common(12): print("  %-28s -> %-28s %8.1f us" % (a, b, g_))
PY
