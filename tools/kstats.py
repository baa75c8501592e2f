"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV (development aid).  usage: kstats.py <kernel_trace.csv> [min_count]"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
print("%-70s %7s %9s %9s %9s %7s" % ("kernel", "calls", "avg us", "median", "total ms", "%"))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < (int(sys.argv[2]) if len(sys.argv) > 2 else 1): continue
    print("%-70s %7d %9.2f %9.2f %9.3f %7.1f" % (k[:70], len(v), sum(v) / len(v), statistics.median(v), sum(v) / 1e3, 100 * sum(v) / tot))
