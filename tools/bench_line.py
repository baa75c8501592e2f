"""Prints a few fields of a bench.py JSON line read from stdin: python bench.py ... | python tools/bench_line.py <label>"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms_per_step", d["ms_per_step"], "min", d.get("ms_per_step_min"), "value", d["value"],
      "cg", d.get("cg_iterations_in_solver_state"), "cost", d.get("final_cost"), "cg_pair_us", r.get("cg_iteration_us_in_situ"))
