#!/bin/bash
# kernel times of one configuration's kernel-pipeline loop (rocprofv3 --kernel-trace --stats): tools/exp/trace_config.sh 3|4s|4k12|4 [TAG]
cd "$(dirname "$0")/../.." && root=$PWD && mkdir -p gpurun_out/r4 && out=$root/gpurun_out/r4/trace_config$1${2:+_$2}.txt && : > $out
export TMPDIR=/tmp
d=/tmp/prof_trace_$1; rm -rf $d
timeout ${T:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python tools/profile_config.py $1 1000000 8 > /tmp/run_trace_$1.txt 2>&1
grep -E "configs|kernel pipeline|packed host|queues" /tmp/run_trace_$1.txt >> $out || tail -5 /tmp/run_trace_$1.txt >> $out
f=$(find $d -name '*kernel_stats.csv' | head -1)
python - "$f" >> $out <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'gmx' in n and float(r['AverageNs']) > 20000 and int(r['Calls']) >= 8:
        print('   %-60s %8.1f us x %s' % (n.split('(')[0].replace('void ','')[:60], float(r['AverageNs'])/1000, r['Calls']))
PY
cat $out
