#!/bin/bash
# A/B of the screening side table (round 6) on one box: configs[4] at an eighth of its length (4s: 400 Mb + 11 M sites, k = 14) or
# at full size (4), kernel pipeline + packed host feed, with and without GMX_NO_SEED_SIDE; per-kernel times under rocprofv3.
#   gpurun --timeout 1500 -- 'bash tools/side_ab.sh 4s'
set -u
export TMPDIR=/tmp
C=${1:-4s}
OUT=gpurun_out/r6/side_ab_$C
mkdir -p $OUT
for mode in side noside; do
  if [ $mode = noside ]; then export GMX_NO_SEED_SIDE=1; else unset GMX_NO_SEED_SIDE; fi
  rm -rf $OUT/trace_$mode
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o trace -- python tools/profile_config.py $C 1000000 6 > $OUT/run_$mode.txt 2>&1
  find $OUT/trace_$mode -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$mode.csv \;
  rm -rf $OUT/trace_$mode
  echo "== $mode"; grep -E "configs|kernel pipeline|packed host feed|stats" $OUT/run_$mode.txt | cut -c1-300
  head -8 $OUT/kernel_stats_$mode.csv | cut -c1-160
done
