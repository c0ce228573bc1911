#!/bin/bash
# kernel trace of the nested-PRG scale check.  gpurun --timeout 900 -- 'bash tools/profile_nested.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_nested
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nested -o trace -- python tools/scale_check_nested.py 300 6000 10 500000 20000 > gpurun_out/prof_nested.log 2>&1
tail -2 gpurun_out/prof_nested.log | cut -c1-200
find gpurun_out/prof_nested -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-150 | head -14
