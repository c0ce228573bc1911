#!/bin/bash
# The round-6 profiles (copy what is to be judged into profiles/round6/):  gpurun --timeout 3000 -- 'bash tools/profile_round6.sh'
#   1. bench.py as the driver runs it (configs[4] included on a box with >= 280 GiB)  -> bench.log (the JSON line)
#   2. rocprofv3 --kernel-trace --stats of the same command without side legs         -> bench_kernel_stats.csv, step timeline
#   3. FETCH_SIZE / WRITE_SIZE passes (own passes, never with a trace)                -> hbm_traffic.json
#   4. SQ counters (two passes) + the stats build's loop statistics                   -> sq_counters.txt, loop_stats.txt, sq_extend.json
#   5. configs[2], configs[3]: kernel stats, FETCH / WRITE, loop statistics           -> config{2,3}/..., config{2,3}_roofline.json
# PASSES="bench trace hbm sq configs" selects parts.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r6
PASSES=${PASSES:-"bench trace hbm sq configs"}
mkdir -p $OUT
STATS=$PWD/gramtools_amd/lib/libgmx_stats.so
has() { case " $PASSES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
fi
if has trace; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
  cp $OUT/trace/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
  python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv gmx_batch_begin > $OUT/step_timeline_host_feed.txt 2>&1
  rm -rf $OUT/trace
fi
if has hbm; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
    echo "pmc $c rc=$?"
  done
  python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
fi
if has sq; then
  run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
  run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
  python tools/pmc_summary.py $OUT/pmc_sq1/pmc_counter_collection.csv $OUT/pmc_sq2/pmc_counter_collection.csv > $OUT/sq_counters.txt
  rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2
  if [ -f $STATS ]; then GMX_LIB=$STATS python tools/loop_stats.py > $OUT/loop_stats.txt 2> $OUT/loop_stats.err; python tools/sq_extend.py $OUT > $OUT/sq_extend.json 2> $OUT/sq_extend.err; fi
fi
if has configs; then
  for C in 2 3; do
    D=$OUT/config$C; mkdir -p $D
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o trace -- python tools/profile_config.py $C 1000000 6 > $D/run_trace.txt 2>&1
    find $D/trace -name '*kernel_stats.csv' -exec cp {} $D/kernel_stats.csv \;
    rm -rf $D/trace
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $c --output-format csv -d $D/$c -o pmc -- python tools/profile_config.py $C 1000000 3 > $D/run_$c.txt 2>&1
      echo "config $C pmc $c rc=$?"
    done
    python tools/hbm_traffic.py $(find $D/FETCH_SIZE -name '*counter_collection.csv') $(find $D/WRITE_SIZE -name '*counter_collection.csv') > $D/hbm_traffic.json
    rm -rf $D/FETCH_SIZE $D/WRITE_SIZE
    if [ -f $STATS ]; then GMX_LIB=$STATS timeout 600 python tools/profile_config.py $C 1000000 2 > $D/loop_stats.txt 2>&1; fi
    python tools/config_roofline.py $C $D > $OUT/config${C}_roofline.json 2> $D/roofline.err
    grep -E "kernel pipeline|packed host" $D/run_trace.txt
  done
fi
tail -1 $OUT/bench.log 2>/dev/null | cut -c1-300
