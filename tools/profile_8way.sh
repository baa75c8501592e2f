# tools/profile_8way.sh <tag> — run ON THE GPU BOX: BASELINE configs[3] (100 k poses / 1 M edges) row-sharded over EIGHT loopback ranks of
# this one GPU (one host thread, one problem, one stream per rank; the same ownership rule, kernels and exchange points as over RCCL):
# rocprofv3 kernel trace and FETCH_SIZE / WRITE_SIZE passes of their own -> what ONE RANK's kernels take and move per launch
# (profiles/<tag>_c4_8way_kernel_stats.csv, <tag>_c4_8way_pmc.json, <tag>_c4_8way_summary.json).  Not a scaling measurement: the eight
# ranks share one device.
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/shard_pipe_check.py 100000 1000000 8 6"
export PGO_SHARD_TRACE_ONLY=1 PGO_SHARD_PIPE=1
rocprofv3 --kernel-trace -d $OUT/s8 -o t -- $CMD > $OUT/c4_8way.log 2>&1
python tools/rocprof_summary.py $OUT/s8/t_results.db $OUT/${TAG}_c4_8way_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/s8f -o f -- $CMD > /dev/null 2> $OUT/c4_8way_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/s8w -o w -- $CMD > /dev/null 2> $OUT/c4_8way_write.err
python tools/rocprof_pmc.py $OUT/s8f/f_results.db $OUT/s8w/w_results.db $OUT/${TAG}_c4_8way_pmc.json > $OUT/c4_8way_pmc.log 2>&1
rm -rf $OUT/s8 $OUT/s8f $OUT/s8w
python - <<PY
import csv, json
N, E, W, CL = 100000, 1000000, 8, 2
alg = {"k_pipe_cg_sym": ((N + E) * 288 + 2 * N * 48 + 10 * N * 48 + N * 36 * CL * 8) / W, "k_linearize_lean<3>": (640 * E + 392 * N) / W}
pm = json.load(open("$OUT/${TAG}_c4_8way_pmc.json"))["kernels"]
rows = [r for r in csv.reader(l for l in open("$OUT/${TAG}_c4_8way_kernel_stats.csv") if not l.startswith("#"))][1:]
out = {"what": "BASELINE configs[3] sharded over 8 loopback ranks of ONE GPU: per-rank launches (rocprofv3 median / PMC median of the active launches) against the rank's share of the algorithmic bytes (SURVEY 8d figure / 8)", "kernels": {}}
for key, a in alg.items():
    base = key.split("<")[0]
    row = next((r for r in rows if base in r[0] and "[" not in r[0]), None)
    p = pm.get(key) or pm.get(base)
    if row is None or p is None: continue
    med = float(row[4])
    out["kernels"][key] = {"launches": int(row[1]), "median_us": med, "algorithmic_bytes_per_rank_launch": int(a), "hbm_bytes_per_launch_pmc": p["hbm_bytes_per_launch_corrected"],
                           "traffic_over_algorithmic": round(p["hbm_bytes_per_launch_corrected"] / a, 3),
                           "note": "eight ranks' launches overlap on one device: the duration of a launch is not a one-rank-per-GPU figure"}
json.dump(out, open("$OUT/${TAG}_c4_8way_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -3 $OUT/c4_8way.log
