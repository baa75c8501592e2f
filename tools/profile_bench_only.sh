# tools/profile_bench_only.sh <tag> — steps 1 and 2 of tools/profile_round.sh alone (kernel trace of the bench command with the one
# universal-stream symbol split by operation; FETCH_SIZE / WRITE_SIZE passes), for when only the PCG kernels changed.  Run ON THE GPU BOX.
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-c4-kernels --cpu-iters 0 --no-cpu-baseline --no-exact-blocks --repeats 2"
rm -f $OUT/oplog.txt
PGO_UNI_OPLOG=$OUT/oplog.txt rocprofv3 --kernel-trace -d $OUT/bench_trace -o bench -- $BENCH --no-quality > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/bench_trace.err
python tools/rocprof_summary.py $OUT/bench_trace/bench_results.db $OUT/${TAG}_bench_kernel_stats.csv $OUT/oplog.txt > /dev/null
rm -rf $OUT/bench_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $BENCH --no-quality > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $BENCH --no-quality > /dev/null 2> $OUT/pmc_write.err
python tools/rocprof_pmc.py $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db $OUT/${TAG}_pmc.json > $OUT/pmc.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/${TAG}_pmc.json $OUT/${TAG}_bench_kernel_stats.csv $OUT/${TAG}_bench_under_rocprof.json profiles/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
