#!/bin/bash
# Every profile DESIGN.md / bench.py cite for this round, written under gpurun_out/prof_r2 (copy what is to be judged into
# profiles/round2/):  gpurun --timeout 2400 -- 'bash tools/profile_round2.sh'
#   1. bench.py as the driver runs it                         -> bench.log
#   2. rocprofv3 --kernel-trace --stats of the same command   -> bench_kernel_stats.csv, step_timeline.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)   -> hbm_traffic.json
#   4. rocprofv3 --pmc SQ counters (own passes)               -> sq_counters.txt
#   5. configs[1] with 0.2 % / 5 % of the genome in 10-copy repeats -> repeat_timeline_*.txt
#   6. loop statistics from a -DGMX_LOOP_STATS build (LAST: it replaces libgmx.so in this scratch copy)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/trace/trace_kernel_trace.csv   # tens of MB: the timeline and the stats are what is kept
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
python tools/pmc_summary.py $OUT/pmc_sq1/pmc_counter_collection.csv $OUT/pmc_sq2/pmc_counter_collection.csv > $OUT/sq_counters.txt
rm -rf $OUT/pmc_*/pmc_counter_collection.csv $OUT/pmc_*/*agent_info.csv
for F in 0.002 0.05; do bash tools/repeat_timeline.sh $F $OUT/rep 2>&1 | grep -v "^[EW]2026" > $OUT/repeat_timeline_$F.txt; done   # configs[1] with repeats: rate + one step's kernels
rm -rf $OUT/rep/trace_*
bash tools/loop_stats.sh > $OUT/loop_stats.log 2>&1; cp gpurun_out/loop_stats.txt $OUT/loop_stats.txt 2>/dev/null
python tools/sq_extend.py $OUT > $OUT/sq_extend.json 2>$OUT/sq_extend.err
cat $OUT/sq_extend.json; tail -1 $OUT/bench.log | cut -c1-400
