#!/bin/bash
# configs[2] (nested): rate + kernel stats + one step's timeline.  gpurun -- 'bash tools/exp/config2_timeline.sh OUT'
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/c2}; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o trace -- python tools/scale_check_configs.py 2 1000000 > $OUT/config2_run.txt 2>&1
grep -E "configs|index:|device-resident|queues" $OUT/config2_run.txt | cut -c1-400
python tools/step_timeline.py $OUT/c2/trace_kernel_trace.csv pack > $OUT/config2_step_timeline.txt 2>&1
head -50 $OUT/config2_step_timeline.txt | cut -c1-130
rm -rf $OUT/c2
