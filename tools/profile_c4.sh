# tools/profile_c4.sh <tag> — run ON THE GPU BOX: rocprofv3 evidence at BASELINE configs[3] size (100 k poses / 1 M edges, one GPU),
# the size SURVEY 8d says the HBM fractions are meaningful at.  Kernel trace in one pass, FETCH_SIZE / WRITE_SIZE in passes of their own.
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/c4_profile.py 8"
rocprofv3 --kernel-trace -d $OUT/c4_trace -o c4 -- $CMD > $OUT/c4_trace.log 2>&1
python tools/rocprof_summary.py $OUT/c4_trace/c4_results.db $OUT/${TAG}_c4_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/c4_fetch -o f -- $CMD > /dev/null 2> $OUT/c4_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/c4_write -o w -- $CMD > /dev/null 2> $OUT/c4_write.err
python tools/rocprof_pmc.py $OUT/c4_fetch/f_results.db $OUT/c4_write/w_results.db $OUT/${TAG}_c4_pmc.json > $OUT/c4_pmc.log 2>&1
rm -rf $OUT/c4_trace $OUT/c4_fetch $OUT/c4_write
tail -n 4 $OUT/c4_trace.log; head -8 $OUT/${TAG}_c4_kernel_stats.csv | cut -c1-160; cat $OUT/c4_pmc.log
