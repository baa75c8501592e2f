"""Per-kernel FETCH_SIZE / WRITE_SIZE (KiB per dispatch) from two rocprofv3 --pmc passes (rocpd databases),
written as profiles/<tag>_pmc.json.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced streams, so the read side is doubled; WRITE_SIZE is reported as counted.
usage: python tools/rocprof_pmc.py <fetch.db> <write.db> <out.json>"""
import hashlib
import json
import os
import re
import sqlite3
import statistics
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)).fetchall()
    d = {}
    for n, v in rows:
        m = re.search(r"(k_[a-z_]+)(?:<(\d+)[,>])?", n)      # kernel + its first template argument: k_spmv<0, true> -> k_spmv<0>
        key = (m.group(1) + ("<%s>" % m.group(2) if m.group(2) is not None else "")) if m else n
        d.setdefault(key, []).append(v)
    return d


def main(fetch_db, write_db, out):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        # early-exit launches of the CG kernels move almost nothing: take the upper half for "active" launches
        fa = sorted(fv)[len(fv) // 2:]
        wa = sorted(wv)[len(wv) // 2:]
        res[k] = {"dispatches": len(fv), "fetch_kib_median_active": statistics.median(fa), "write_kib_median_active": statistics.median(wa),
                  "hbm_bytes_per_launch_corrected": int(2 * statistics.median(fa) * 1024 + statistics.median(wa) * 1024),
                  "hbm_bytes_per_launch_raw": int(statistics.median(fa) * 1024 + statistics.median(wa) * 1024)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import pgo_loader
    sha = pgo_loader.load().kernel_source_sha()      # (as bench.py computes it)
    json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated",
               "kernel_source_sha256_16": sha, "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-22s n=%5d fetch %10.1f KiB write %10.1f KiB -> corrected %.2f MB" % (k, v["dispatches"], v["fetch_kib_median_active"], v["write_kib_median_active"], v["hbm_bytes_per_launch_corrected"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
