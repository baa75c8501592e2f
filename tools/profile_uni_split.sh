cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04a; mkdir -p $OUT
BENCH="python bench.py --no-c4-kernels --cpu-iters 0 --no-cpu-baseline --no-exact-blocks --repeats 2"
rm -f $OUT/oplog.txt
PGO_UNI_OPLOG=$OUT/oplog.txt rocprofv3 --kernel-trace -d $OUT/bench_trace -o bench -- $BENCH > $OUT/bench_trace.json 2> $OUT/bench_trace.err
python tools/rocprof_summary.py $OUT/bench_trace/bench_results.db $OUT/r04a_bench_kernel_stats.csv $OUT/oplog.txt > /dev/null
head -12 $OUT/r04a_bench_kernel_stats.csv
rm -rf $OUT/bench_trace
$BENCH > $OUT/bench_noprof.json 2>/dev/null
