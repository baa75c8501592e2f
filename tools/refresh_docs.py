"""Development aid: after `tools/profile_round.sh rNN` and `cp gpurun_out/rNN/rNN_* profiles/` + the bench line — rewrites the figures DESIGN.md /
README.md / INTEGRATION.md quote from the committed profiles (tests/test_docs_consistency.py ties them) from the PREVIOUS commit's values to the new
files' values.  usage (repo root, before committing the new profiles): python tools/refresh_docs.py [tag]"""
import csv
import json
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def old(path):
    return subprocess.check_output(["git", "show", "HEAD:" + path]).decode()


def rows(text):
    return [r for r in csv.reader(l for l in text.split("\n") if l and not l.startswith("#"))][1:]


def cg(text):
    r = rows(text)
    return ([x for x in r if x[0].endswith("[cg]")] or [x for x in r if "k_res_cg" in x[0]])[0]


def thousands(x):
    s = str(int(x))
    return s[:-3] + " " + s[-3:] if len(s) > 3 else s


design, readme, integ = open("DESIGN.md").read(), open("README.md").read(), open("INTEGRATION.md").read()
tab = open("profiles/%s_config_table.md" % tag).read().strip("\n")
design = re.sub(r"(<!-- config-table:begin[^>]*-->\n)(.*?)(\n<!-- config-table:end -->)", lambda m: m.group(1) + tab + m.group(3), design, flags=re.S)
o, n = cg(old("profiles/%s_bench_kernel_stats.csv" % tag)), cg(open("profiles/%s_bench_kernel_stats.csv" % tag).read())
ob, nb = json.loads(old("profiles/%s_bench.json" % tag)), json.load(open("profiles/%s_bench.json" % tag))
orr, nr = ob["roofline"], nb["roofline"]
oq, nq = ob["c2_solution_quality"], nb["c2_solution_quality"]
pairs = [
    ("**%.2f µs average** over %s launches" % (float(o[3]), thousands(o[1])), "**%.2f µs average** over %s launches" % (float(n[3]), thousands(n[1]))),
    ("396 MB / %.2f µs = %.2f TB/s" % (float(o[3]), 396.0576 / float(o[3])), "396 MB / %.2f µs = %.2f TB/s" % (float(n[3]), 396.0576 / float(n[3]))),
    ("device clock of the traced CG launches (%.1f µs) → %.4f." % (orr["avg_launch_us"], orr["frac"]), "device clock of the traced CG launches (%.1f µs) → %.4f." % (nr["avg_launch_us"], nr["frac"])),
    ("**%.4f** at 25 steps (the box of the committed" % ob["ms_per_step"], "**%.4f** at 25 steps (the box of the committed" % nb["ms_per_step"]),
    ("%.2f µs per turn (the tail's" % ob["us_per_cg_turn"], "%.2f µs per turn (the tail's" % nb["us_per_cg_turn"]),
    ("%.1f µs fixed." % ob["fixed_us_per_lm_iteration"], "%.1f µs fixed." % nb["fixed_us_per_lm_iteration"]),
    ("`lm_iteration_ms_one_gpu` there: %.2f on the committed box" % orr["at_c4_size"]["lm_iteration_ms_one_gpu"], "`lm_iteration_ms_one_gpu` there: %.2f on the committed box" % nr["at_c4_size"]["lm_iteration_ms_one_gpu"]),
]
for key, a64 in (("eta_0.1_aggregates_of_64", True), ("eta_0.1_aggregates_of_128", False)):
    a, b = oq["pcg_with_coarse_level"][key], nq["pcg_with_coarse_level"][key]
    fmt = "| %.3f s | **%.3f s** |" if a64 else "| %.3f s | %.3f s |"
    pairs.append((fmt % (a["wall_seconds"], a["seconds_to_target"]), fmt % (b["wall_seconds"], b["seconds_to_target"])))
a, b = oq["pcg"]["eta_1e-05"], nq["pcg"]["eta_1e-05"]
pairs.append(("| %.2f s | %.3f s |" % (a["wall_seconds"], a["seconds_to_target"]), "| %.2f s | %.3f s |" % (b["wall_seconds"], b["seconds_to_target"])))
oc4, nc4 = rows(old("profiles/%s_c4_kernel_stats.csv" % tag)), rows(open("profiles/%s_c4_kernel_stats.csv" % tag).read())
op, npm = json.loads(old("profiles/%s_c4_pmc.json" % tag)), json.load(open("profiles/%s_c4_pmc.json" % tag))
for key, alg, lead in (("k_spmv_sym<0", 326.4, "C4: **%s µs → %.2f of 8 TB/s**"), ("k_linearize_lean<3", 679.2, "C4 **%s µs → %.2f of 8 TB/s**")):
    a = "%.1f" % float([r for r in oc4 if key in r[0]][0][4])
    b = "%.1f" % float([r for r in nc4 if key in r[0]][0][4])
    pairs.append((lead % (a, alg / float(a) / 8), lead % (b, alg / float(b) / 8)))
    pa = "%.1f" % (op["kernels"][key + ">"]["hbm_bytes_per_launch_corrected"] / 1e6)
    pb = "%.1f" % (npm["kernels"][key + ">"]["hbm_bytes_per_launch_corrected"] / 1e6)
    pairs.append((pa, pb))
for a, b in pairs:
    if a == b:
        continue
    if a not in design:
        print("not found in DESIGN.md (edit by hand):", a, "->", b)
    design = design.replace(a, b)
c_o, c_n = oq["pcg_with_coarse_level"]["eta_0.1_aggregates_of_64"], nq["pcg_with_coarse_level"]["eta_0.1_aggregates_of_64"]
e_o, e_n = oq["pcg"]["eta_1e-05"]["seconds_to_target"], nq["pcg"]["eta_1e-05"]["seconds_to_target"]
readme = readme.replace("(`profiles/%s_bench.json`: %.4f;" % (tag, ob["ms_per_step"]), "(`profiles/%s_bench.json`: %.4f;" % (tag, nb["ms_per_step"]))
readme = readme.replace("%.2f s of wall, and\n  passes `exact cost × (1 + 1e-3)` after **%.0f ms** — against %.0f ms for `η = 1e-5`" % (c_o["wall_seconds"], 1e3 * c_o["seconds_to_target"], 1e3 * e_o),
                        "%.2f s of wall, and\n  passes `exact cost × (1 + 1e-3)` after **%.0f ms** — against %.0f ms for `η = 1e-5`" % (c_n["wall_seconds"], 1e3 * c_n["seconds_to_target"], 1e3 * e_n))
integ = integ.replace("after %.2f s, and passes the exact steps' cost after %.0f ms" % (c_o["wall_seconds"], 1e3 * c_o["seconds_to_target"]),
                      "after %.2f s, and passes the exact steps' cost after %.0f ms" % (c_n["wall_seconds"], 1e3 * c_n["seconds_to_target"]))
open("DESIGN.md", "w").write(design)
open("README.md", "w").write(readme)
open("INTEGRATION.md", "w").write(integ)
print("ms_per_step %.4f -> %.4f; check `git diff` and run tests/test_docs_consistency.py" % (ob["ms_per_step"], nb["ms_per_step"]))
