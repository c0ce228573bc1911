#!/bin/bash
# gmx_cover_jump_kernel: geometry records fetched side by side (GMX_JUMP_BATCH) against waves per SIMD (GMX_JUMP_MIN_BLOCKS), configs[3]
cd "$(dirname "$0")/../.." && root=$PWD && mkdir -p gpurun_out/r4 && out=$root/gpurun_out/r4/jump_variants.txt && : > $out
export TMPDIR=/tmp
for v in A B C D E; do
  d=/tmp/prof_v$v; rm -rf $d
  echo "== variant $v" >> $out
  GMX_LIB=$root/gramtools_amd/lib/libgmx_exp$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python tools/profile_config.py 3 1000000 8 > /tmp/run_v$v.txt 2>&1
  grep -E "kernel pipeline" /tmp/run_v$v.txt >> $out || tail -5 /tmp/run_v$v.txt >> $out
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" >> $out <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('cover_jump','cover_one','cover_coop_kernel<3>')):
        print('   %-50s %8.1f us x %s' % (n.split('(')[0][:50], float(r['AverageNs'])/1000, r['Calls']))
PY
done
cat $out
