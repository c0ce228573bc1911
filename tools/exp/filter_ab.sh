#!/bin/bash
# A/B of the k-mer filter forms on one box: kernel pipeline (tools/kbench.py) + step timeline, with the variable given set / unset.
# Usage: gpurun -- 'bash tools/exp/filter_ab.sh OUT VAR=VALUE [VAR=VALUE ...]'   (each assignment is one more leg beside the default)
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/filter_ab}; shift; rm -rf $OUT; mkdir -p $OUT
leg() {  # name, env assignments...
  local name=$1; shift
  echo "== $name" | tee -a $OUT/ab.txt
  env "$@" python tools/kbench.py 2>&1 | cut -c1-330 | tee -a $OUT/ab.txt
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$name -o trace -- python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $OUT/prof_$name.log 2>&1
  python tools/step_timeline.py $OUT/t_$name/trace_kernel_trace.csv > $OUT/timeline_$name.txt 2>&1
  grep -E "filter|extend_kernel<false, 1>|cover_jump" $OUT/t_$name/trace_kernel_stats.csv | cut -c1-140 | tee -a $OUT/ab.txt
  rm -rf $OUT/t_$name
}
leg default GMX_DUMMY=1
i=0
for a in "$@"; do i=$((i+1)); leg "v$i" $(echo $a | tr ',' ' '); done
