#!/bin/bash
# kernel trace of a scale check.  gpurun --timeout 1500 -- 'bash tools/profile_scale.sh 64444167 1800000 14 4000000'
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_scale
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scale -o trace -- python tools/scale_check.py "$@" > gpurun_out/prof_scale.log 2>&1
tail -2 gpurun_out/prof_scale.log | cut -c1-200
find gpurun_out/prof_scale -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-150 | head -16
