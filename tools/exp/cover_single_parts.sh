#!/bin/bash
# what gmx_cover_single_kernel spends its time on at configs[3]: builds with parts of it cut out (GMX_EXP bits: 1 = no atomics,
# 2 = walk tasks skipped, 4 = walk-free tasks skipped), kernel times from rocprofv3 --kernel-trace --stats
cd "$(dirname "$0")/../.." && root=$PWD && mkdir -p gpurun_out/r4 && out=$root/gpurun_out/r4/cover_single_parts.txt && : > $out
export TMPDIR=/tmp
for v in 0 1 2 4 3; do
  lib=$root/gramtools_amd/lib/libgmx.so; [ $v != 0 ] && lib=$root/gramtools_amd/lib/libgmx_exp$v.so
  d=/tmp/prof_exp$v; rm -rf $d
  echo "== GMX_EXP=$v" >> $out
  GMX_LIB=$lib timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python tools/profile_config.py 3 1000000 8 > /tmp/run_exp$v.txt 2>&1
  grep -E "kernel pipeline" /tmp/run_exp$v.txt >> $out || tail -5 /tmp/run_exp$v.txt >> $out
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" >> $out <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('cover_single','cover_one','cover_coop_kernel<3>','extend_kernel')):
        print('   %-50s %8.1f us x %s' % (n.split('(')[0][:50], float(r['AverageNs'])/1000, r['Calls']))
PY
done
cat $out
