"""Per-kernel FP64 MFMA counters from a rocprofv3 --pmc pass (rocpd database) -> profiles/<tag>_front_c5_mfma_pmc.json.
SQ_INSTS_VALU_MFMA_MOPS_F64 counts MFMA work in units of 512 flops (rocprofv3's own MfmaFlopsF64 = MOPS * 512);
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) is rocprofv3's MfmaUtil.  usage: rocprof_mfma.py <db> <out.json>"""
import json
import re
import sqlite3
import sys

SIMDS = 256 * 4


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    per = {}
    for name, counter, value, disp in rows:
        m = re.search(r"(k_[a-z_0-9]+(<\d+>)?)", name)
        key = m.group(1) if m else name
        per.setdefault(key, {}).setdefault(counter, []).append(value)
    dur = {}
    for name, d in c.execute("select name, duration from kernels").fetchall():
        m = re.search(r"(k_[a-z_0-9]+(<\d+>)?)", name)
        dur[m.group(1) if m else name] = dur.get(m.group(1) if m else name, 0.0) + d * 1e-9
    res = {}
    for k, cs in sorted(per.items()):
        n = max(len(v) for v in cs.values())
        mops = sum(cs.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [0.0]))
        busy = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
        active = sum(cs.get("GRBM_GUI_ACTIVE", [0.0]))
        res[k] = {"dispatches": n, "mfma_mops_f64": mops, "mfma_flops_f64": mops * 512.0, "mfma_busy_cycles": busy,
                  "gui_active_cycles_summed_over_xcds": active, "kernel_seconds": dur.get(k),
                  "tflops_from_counters": (mops * 512.0 / dur[k] / 1e12) if dur.get(k) else None,
                  # busy cycles over (kernel time x 2.4 GHz x 1024 SIMDs); rocprofv3's MfmaUtil divides by GRBM_GUI_ACTIVE, which this
                  # chip reports summed over its 8 XCDs (the derived metric comes out 8x too small)
                  "mfma_util_percent": (100.0 * busy / (dur[k] * 2.4e9 * SIMDS)) if dur.get(k) else None}
    total_flops = sum(v["mfma_flops_f64"] for v in res.values())
    json.dump({"note": "FP64 MFMA counters, summed over all dispatches of each kernel in the run (tools/front_prof.py c5 3: 5 factorisations)",
               "total_mfma_flops_f64": total_flops, "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        if v["mfma_mops_f64"]:
            print("%-24s n=%5d MFMA flops %.4e in %.4f s = %.2f TFLOP/s, busy %.3e cycles, MfmaUtil %.2f %%" % (
                k, v["dispatches"], v["mfma_flops_f64"], v["kernel_seconds"] or 0.0, v["tflops_from_counters"] or 0.0, v["mfma_busy_cycles"], v["mfma_util_percent"] or 0.0))
    print("total FP64 MFMA flops %.4e" % total_flops)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
