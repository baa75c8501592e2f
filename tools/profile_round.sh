# tools/profile_round.sh <tag> — run ON THE GPU BOX (gpurun): collects the rocprofv3 evidence of a round under gpurun_out/<tag>/
# and writes the summaries that get committed under profiles/<tag>_*.  Counters are collected in passes of their own
# (--pmc without any trace domain besides the kernel trace), as the pool requires.
#   1. kernel trace of the bench command (C2 PCG timed region + exact blocks), per-kernel summary
#   2. FETCH_SIZE / WRITE_SIZE passes of the same command -> HBM bytes per launch (tools/rocprof_pmc.py)
#   3. kernel trace of the multifrontal factorisation on C2 and C5
#   4. FP64 MFMA counters of the multifrontal factorisation on C5
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-c4-kernels --cpu-iters 0 --no-cpu-baseline --no-exact-blocks --repeats 2"
# (PGO_UNI_OPLOG: the library logs what every k_uni_s launch did, so that the summary can split that one symbol by operation;
# the bench line printed UNDER the profiler is kept: its HIP-event figures are the ones comparable with the trace)
rm -f $OUT/oplog.txt
PGO_UNI_OPLOG=$OUT/oplog.txt rocprofv3 --kernel-trace -d $OUT/bench_trace -o bench -- $BENCH --no-quality > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/bench_trace.err
python tools/rocprof_summary.py $OUT/bench_trace/bench_results.db $OUT/${TAG}_bench_kernel_stats.csv $OUT/oplog.txt > /dev/null
rm -rf $OUT/bench_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $BENCH --no-quality > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $BENCH --no-quality > /dev/null 2> $OUT/pmc_write.err
python tools/rocprof_pmc.py $OUT/pmc_fetch/f_results.db $OUT/pmc_write/w_results.db $OUT/${TAG}_pmc.json > $OUT/pmc.log 2>&1
for c in c2 c5; do
  rocprofv3 --kernel-trace -d $OUT/front_$c -o front -- python tools/front_prof.py $c 5 > $OUT/front_$c.log 2>&1
  python tools/rocprof_summary.py $OUT/front_$c/front_results.db $OUT/${TAG}_front_${c}_kernel_stats.csv
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/front_mfma -o m -- python tools/front_prof.py c5 3 > $OUT/front_mfma.log 2>&1
python tools/rocprof_mfma.py $OUT/front_mfma/m_results.db $OUT/${TAG}_front_c5_mfma_pmc.json > $OUT/front_mfma_summary.log 2>&1
# 5. the sharded path on eight loopback ranks (C4), both CG forms; the one-rank all-gather timing; the FP64 MFMA clock measurement
for m in 1 0; do
  PGO_SHARD_TRACE_ONLY=1 PGO_SHARD_PIPE=$m rocprofv3 --kernel-trace -d $OUT/sh$m -o t -- python tools/shard_pipe_check.py 100000 1000000 8 6 > $OUT/shard_log$m.txt 2>&1
  python tools/rocprof_summary.py $OUT/sh$m/t_results.db $OUT/${TAG}_c4_8way_loopback_pipe${m}_kernel_stats.csv > /dev/null
  rm -rf $OUT/sh$m
done
# ... and with the exchange done by the kernels themselves (DeviceGraph::peer_tab; every virtual rank's stream on its own hardware queue)
GPU_MAX_HW_QUEUES=16 PGO_PEER_DIRECT=1 PGO_SHARD_TRACE_ONLY=1 PGO_SHARD_PIPE=1 rocprofv3 --kernel-trace -d $OUT/shd -o t -- python tools/shard_pipe_check.py 100000 1000000 8 6 > $OUT/shard_log_direct.txt 2>&1
python tools/rocprof_summary.py $OUT/shd/t_results.db $OUT/${TAG}_c4_8way_loopback_direct_kernel_stats.csv > /dev/null
rm -rf $OUT/shd
python tools/exchange_latency.py 2>/dev/null | grep "^{" > $OUT/exchange_world1_raw.json
[ -x tools/bench/mfma_clock ] && tools/bench/mfma_clock > $OUT/${TAG}_mfma_clock.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/front_c2 $OUT/front_c5 $OUT/front_mfma
bash tools/profile_c4.sh $TAG > $OUT/profile_c4.log 2>&1
# 6. (r06) the eight-way sharded run once more with FETCH_SIZE / WRITE_SIZE passes: what one rank's kernels move per launch
bash tools/profile_8way.sh $TAG > $OUT/profile_8way.log 2>&1
python tools/config_table.py > $OUT/${TAG}_config_table.md 2> $OUT/config_table.err
cp $OUT/${TAG}_pmc.json $OUT/${TAG}_bench_kernel_stats.csv $OUT/${TAG}_bench_under_rocprof.json profiles/ 2>/dev/null   # on this box only: the final bench run quotes the traffic measured above, on these very sources
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -n 3 $OUT/pmc.log $OUT/front_mfma_summary.log $OUT/front_c2.log $OUT/front_c5.log
# then, in the build container: cp gpurun_out/<tag>/<tag>_* profiles/ (gpurun merges only gpurun_out/ back)
