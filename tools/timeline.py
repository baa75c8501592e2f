"""Kernel timeline of ONE LM iteration from a rocprofv3 kernel-trace database (rocpd): start offset, duration and the idle
gap before each kernel, between two consecutive k_pcg_init launches in the middle of the run.
usage: python tools/timeline.py <results.db> [which_iteration] [anchor kernel, default k_pcg_init; k_linearize for the exact path]"""
import re
import sqlite3
import sys


def main(db, which=None, anchor="k_pcg_init"):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else "start_timestamp"
    ec = "end" if "end" in cols else "end_timestamp"
    rows = c.execute("select name, %s, %s from kernels order by %s" % (sc, ec, sc)).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    k = int(which) if which not in (None, "-") else len(idx) // 2
    a, b = idx[k], idx[k + 1]
    t0 = rows[a][1]
    prev_end = rows[a - 1][2]
    busy = 0.0
    agg = {}
    for n, s, e in rows[a:b]:
        m = re.search(r"(k_[a-z_]+(<\d>)?)", n)
        nm = m.group(1) if m else n[:40]
        print("%-24s +%9.2f us  dur %7.2f  gap %7.2f" % (nm, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += (e - s) / 1e3
        cnt, tot, gp = agg.get(nm, (0, 0.0, 0.0))
        agg[nm] = (cnt + 1, tot + (e - s) / 1e3, gp + (s - prev_end) / 1e3)
        prev_end = e
    span = (rows[b][1] - t0) / 1e3
    print("span %.1f us, kernels busy %.1f us, idle %.1f us" % (span, busy, span - busy))
    for nm, (cnt, tot, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-24s x%3d  %8.1f us busy  %8.1f us gaps-before" % (nm, cnt, tot, gp))


if __name__ == "__main__":
    main(*sys.argv[1:])
