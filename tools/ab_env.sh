#!/bin/bash
# A/B of one environment switch on one box: bench value + per-kernel times (rocprofv3 kernel trace) with and without it.
# Usage: gpurun -- 'bash tools/ab_env.sh GMX_NO_FUSE OUTDIR'
set -u
VAR=$1; OUT=${2:-gpurun_out/ab}; mkdir -p $OUT; export TMPDIR=/tmp
for mode in on off on off; do
  if [ $mode = off ]; then export $VAR=1; else unset $VAR; fi
  python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR unset' if '$mode'=='on' else '$VAR=1', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],4), 'ms/step; extend', round(d['roofline']['avg_launch_ms'],4))" | tee -a $OUT/ab.txt
done
for mode in on off; do
  if [ $mode = off ]; then export $VAR=1; else unset $VAR; fi
  rm -rf $OUT/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o trace -- python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench_prof_$mode.log 2>&1
  echo "== $mode"; find $OUT/prof_$mode -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-150 | head -12 | tee -a $OUT/ab.txt
done
