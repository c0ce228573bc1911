#!/bin/bash
# The bench-line profiles of round 4 (configs[1]), written under gpurun_out/prof_r4 (copy what is to be judged into
# profiles/round4/):  gpurun --timeout 2400 -- 'bash tools/profile_round4_bench.sh'
#   1. bench.py as the driver runs it                               -> bench.log
#   2. rocprofv3 --kernel-trace --stats of the same command         -> bench_kernel_stats.csv, step_timeline_*.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)         -> hbm_traffic.json
#   4. SQ counters (own passes) + loop statistics                   -> sq_counters.txt, loop_stats.txt, sq_extend.json
#   5. configs[2] rate + kernel stats, repeats 0.2 % / 5 %          -> config2_*.txt, repeat_timeline_*.txt
#   6. gzip feeds: BGZF, plain gzip on all threads, zlib alone      -> gz_feed.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r4
rm -rf $OUT; mkdir -p $OUT
STATS=$PWD/gramtools_amd/lib/libgmx_stats.so
python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv pack > $OUT/step_timeline_device_resident.txt 2>&1
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv gmx_batch_begin > $OUT/step_timeline_host_feed.txt 2>&1
rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
python tools/pmc_summary.py $OUT/pmc_sq1/pmc_counter_collection.csv $OUT/pmc_sq2/pmc_counter_collection.csv > $OUT/sq_counters.txt
rm -rf $OUT/pmc_*
if [ -f $STATS ]; then GMX_LIB=$STATS python tools/loop_stats.py > $OUT/loop_stats.txt 2> $OUT/loop_stats.err; python tools/sq_extend.py $OUT > $OUT/sq_extend.json 2> $OUT/sq_extend.err; fi
# ---- configs[2] ----
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o trace -- python tools/scale_check_configs.py 2 1000000 > $OUT/config2_run.txt 2>&1
cp $OUT/c2/trace_kernel_stats.csv $OUT/config2_kernel_stats.csv 2>/dev/null
python tools/step_timeline.py $OUT/c2/trace_kernel_trace.csv pack > $OUT/config2_step_timeline.txt 2>&1
rm -rf $OUT/c2
grep -E "configs|index:|device-resident|queues" $OUT/config2_run.txt | cut -c1-500 > $OUT/config2_rate.txt
# ---- repeats ----
for F in 0.002 0.05; do bash tools/repeat_timeline.sh $F $OUT/rep 2>&1 | grep -v "^[EW]2026" > $OUT/repeat_timeline_$F.txt; done
rm -rf $OUT/rep
# ---- gzip feeds ----
python tools/gz_feed.py > $OUT/gz_feed.txt 2>&1
tail -3 $OUT/bench.log | cut -c1-600
