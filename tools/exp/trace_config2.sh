#!/bin/bash
# step timeline of configs[2] (nested MSA regions): tools/exp/trace_config2.sh [TAG]
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4 && export TMPDIR=/tmp && d=/tmp/c2trace && rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o trace -- python tools/scale_check_configs.py 2 1000000 > /tmp/c2run.txt 2>&1
grep -E "device-resident" /tmp/c2run.txt | cut -c1-100 > gpurun_out/r4/config2_timeline${1:+_$1}.txt
python tools/step_timeline.py $d/trace_kernel_trace.csv pack | cut -c1-110 | head -45 >> gpurun_out/r4/config2_timeline${1:+_$1}.txt
cat gpurun_out/r4/config2_timeline${1:+_$1}.txt
