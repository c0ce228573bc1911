#!/bin/bash
# Round 4: counters where the index lives in HBM. For one configuration (3, 4 or 4s):
#   kernel-trace stats, FETCH_SIZE and WRITE_SIZE (own passes, never with a trace), two SQ passes.
#   gpurun --timeout 3000 -- 'bash tools/profile_round4.sh 3'        (config 4: every pass rebuilds the 160 GB index, ~6 min each)
set -u
export TMPDIR=/tmp
C=${1:-3}
PASSES=${2:-"trace FETCH_SIZE WRITE_SIZE sq1 sq2"}
OUT=gpurun_out/r4/config$C
mkdir -p $OUT
T=${GMX_PROFILE_TIMEOUT:-900}
for p in $PASSES; do
  rm -rf $OUT/$p
  case $p in
    trace) timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python tools/profile_config.py $C > $OUT/run_trace.txt 2>&1 ;;
    sq1) timeout $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -o pmc -- python tools/profile_config.py $C 1000000 3 > $OUT/run_sq1.txt 2>&1 ;;
    sq2) timeout $T rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -o pmc -- python tools/profile_config.py $C 1000000 3 > $OUT/run_sq2.txt 2>&1 ;;
    *) timeout $T rocprofv3 --pmc $p --output-format csv -d $OUT/$p -o pmc -- python tools/profile_config.py $C 1000000 3 > $OUT/run_$p.txt 2>&1 ;;
  esac
  echo "pass $p rc=$?"
done
[ -d $OUT/trace ] && find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
if [ -d $OUT/FETCH_SIZE ] && [ -d $OUT/WRITE_SIZE ]; then
  python tools/hbm_traffic.py $(find $OUT/FETCH_SIZE -name '*counter_collection.csv') $(find $OUT/WRITE_SIZE -name '*counter_collection.csv') > $OUT/hbm_traffic.json
fi
for s in sq1 sq2; do
  [ -d $OUT/$s ] && python tools/pmc_summary.py $(find $OUT/$s -name '*counter_collection.csv') > $OUT/${s}_counters.txt
done
# keep the merge small: the raw traces and counter dumps stay on the box
rm -rf $OUT/trace $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/sq1 $OUT/sq2
tail -4 $OUT/run_trace.txt 2>/dev/null
head -25 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200
