"""Turns a rocprofv3 rocpd database (kernel trace) into the per-kernel summary kept under profiles/.
usage: python tools/rocprof_summary.py <results.db> <out.csv> [<oplog.txt>]

The universal stream runs ONE kernel symbol, k_uni_s, for four different jobs (CG product, tail product, refresh product,
linearisation), chosen on the device at run time, so rocprofv3's per-symbol statistics mix them.  With PGO_UNI_OPLOG=<file> the
library logs what every k_uni_s launch did ("<device tick> <op>" per launch, in launch order per problem); given that file, the
dispatches of k_uni_s (ordered by start time) are matched one to one with the log entries (ordered by tick) and reported as
separate rows  k_uni_s[cg] / [tail] / [refresh] / [linearize] / [nop]  next to the un-split row.  The counts must agree, or the
split is refused."""
import sqlite3
import statistics
import sys

OPS = {0: "nop", 1: "cg", 2: "refresh", 3: "linearize", 4: "tail"}
# the fused stream (r05: k_uni_f, one kernel symbol for everything; its log entries carry 16 + operation)
OPS_F = {16: "idle", 17: "head", 18: "w0", 19: "cg", 20: "tail", 21: "linearize"}
# the resident stream (r05: k_res_v / k_res_cg / k_res_lin in a fixed cycle, each launch acting only if its operation is next; 32 + operation)
OPS_R = {32: "idle", 33: "head", 35: "cg", 36: "tail", 37: "linearize"}


def main(db, out, oplog=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size, start from kernels order by start").fetchall()
    d = {}
    for n, du, gx, wx, vg, sg, lds, st in rows:
        d.setdefault(n, []).append((du, gx, wx, vg, sg, lds))
    note = ""
    if oplog:
        entries = sorted(tuple(int(x) for x in line.split()) for line in open(oplog) if line.strip())
        for sym, names, log in (("k_uni_s", OPS, [e for e in entries if e[1] < 16]), ("k_uni_f", OPS_F, [e for e in entries if 16 <= e[1] < 32]),
                                ("k_res_", OPS_R, [e for e in entries if 32 <= e[1] < 48])):
            uni = [(n, r) for n, *r in rows if sym in n]
            if not uni and not log:
                continue
            if len(log) != len(uni):
                note += "# %s split REFUSED: %d dispatches in the trace, %d entries in the operation log\n" % (sym, len(uni), len(log))
                continue
            for (n, r), (_, op) in zip(uni, log):
                d.setdefault("%s[%s]" % (n, names.get(op, str(op))), []).append(tuple(r[:6]))
            note += "# %s[...] rows: the %d dispatches of %s split by the operation each launch performed (PGO_UNI_OPLOG); they are also counted in the un-split row (pct of the split rows is relative to the same total)\n" % (sym, len(uni), sym)
    total = sum(du for n, v in d.items() if "[" not in n for du, *_ in v)
    with open(out, "w") as f:
        f.write(note)
        f.write("kernel,calls,total_us,avg_us,median_us,min_us,max_us,pct,grid_x,workgroup_x,vgpr,sgpr,lds_bytes\n")
        for n, v in sorted(d.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
            du = [x[0] / 1e3 for x in v]
            f.write('"%s",%d,%.1f,%.3f,%.3f,%.3f,%.3f,%.2f,%d,%d,%d,%d,%d\n' % (
                n, len(du), sum(du), sum(du) / len(du), statistics.median(du), min(du), max(du),
                100.0 * sum(du) * 1e3 / total, v[-1][1], v[-1][2], v[-1][3], v[-1][4], v[-1][5]))
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
