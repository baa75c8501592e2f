"""CPU: tools/rocprof_summary.py on a synthetic rocpd database — the dispatches of the resident stream's kernels (k_res_v / k_res_cg /
k_res_lin, a fixed cycle in which a launch may idle) are split by the operation each launch logged (PGO_UNI_OPLOG, entries 32 + operation),
because `roofline.frac` of the bench line is computed from the `[cg]` row of the committed CSV: an idle launch of the CG role must not
be averaged into it, and a log that does not match the trace must be refused, not guessed at."""
import csv
import importlib.util
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rocprof_summary", os.path.join(ROOT, "tools", "rocprof_summary.py"))
rs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rs)

V = "void pgo::(anonymous namespace)::k_res_v<3, 2, %d>(pgo::DeviceGraph, int, double, double)"
CG = "void pgo::(anonymous namespace)::k_res_cg<true, 2>(pgo::DeviceGraph, pgo::CgParams, int)"
LIN = "void pgo::(anonymous namespace)::k_res_lin_lean<3>(pgo::DeviceGraph, int)"


def _db(path, rows):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size, start)")
    c.executemany("insert into kernels values (?, ?, 100352, 256, 128, 100, 0, ?)", rows)
    c.commit()
    c.close()


def _rows(path):
    return {r[0]: r for r in csv.reader(l for l in open(path) if not l.startswith("#")) if r and r[0] != "kernel"}


def test_resident_dispatches_are_split_by_logged_operation(tmp_path):
    # two LM iterations: HEAD CG TAIL LIN | HEAD CG TAIL (rejected: the LIN launch idles) + an idle CG-role launch of a stopped stream
    seq = [(V % 1, 16000, 33), (CG, 120000, 35), (V % 4, 13000, 36), (LIN, 12000, 37),
           (V % 1, 17000, 33), (CG, 40000, 35), (V % 4, 13500, 36), (LIN, 2000, 32), (V % 1, 2100, 32), (CG, 3000, 32)]
    _db(str(tmp_path / "t.db"), [(n, d, 1000 * i) for i, (n, d, _) in enumerate(seq)])
    (tmp_path / "oplog.txt").write_text("".join("%d %d\n" % (500 + 7 * i, op) for i, (_, _, op) in enumerate(seq)))
    rs.main(str(tmp_path / "t.db"), str(tmp_path / "out.csv"), str(tmp_path / "oplog.txt"))
    r = _rows(str(tmp_path / "out.csv"))
    assert int(r[CG][1]) == 3 and int(r[CG + "[cg]"][1]) == 2 and int(r[CG + "[idle]"][1]) == 1
    assert abs(float(r[CG + "[cg]"][3]) - 80.0) < 1e-9 and abs(float(r[CG][3]) - 163.0 / 3) < 1e-3      # average us: the CG launches alone / all three
    assert int(r[(V % 1) + "[head]"][1]) == 2 and int(r[(V % 1) + "[idle]"][1]) == 1 and int(r[(V % 4) + "[tail]"][1]) == 2
    assert int(r[LIN + "[linearize]"][1]) == 1 and int(r[LIN + "[idle]"][1]) == 1


def test_a_log_that_does_not_match_the_trace_is_refused(tmp_path):
    _db(str(tmp_path / "t.db"), [(CG, 120000, 0), (CG, 40000, 1000)])
    (tmp_path / "oplog.txt").write_text("500 35\n")
    rs.main(str(tmp_path / "t.db"), str(tmp_path / "out.csv"), str(tmp_path / "oplog.txt"))
    text = open(str(tmp_path / "out.csv")).read()
    assert "split REFUSED" in text and "[cg]" not in text
