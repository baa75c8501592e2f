"""Turns a rocprofv3 rocpd database (kernel trace) into the per-kernel summary kept under profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db profiles/r01_bench_kernel_stats.csv"""
import sqlite3
import statistics
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels").fetchall()
    d = {}
    for n, du, gx, wx, vg, sg, lds in rows:
        d.setdefault(n, []).append((du, gx, wx, vg, sg, lds))
    total = sum(du for v in d.values() for du, *_ in v)
    with open(out, "w") as f:
        f.write("kernel,calls,total_us,avg_us,median_us,min_us,max_us,pct,grid_x,workgroup_x,vgpr,sgpr,lds_bytes\n")
        for n, v in sorted(d.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
            du = [x[0] / 1e3 for x in v]
            f.write('"%s",%d,%.1f,%.3f,%.3f,%.3f,%.3f,%.2f,%d,%d,%d,%d,%d\n' % (
                n, len(du), sum(du), sum(du) / len(du), statistics.median(du), min(du), max(du),
                100.0 * sum(du) * 1e3 / total, v[-1][1], v[-1][2], v[-1][3], v[-1][4], v[-1][5]))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
