#!/bin/bash
# gmx_cover_single_kernel with a bounded grid: rate of the configs[3] kernel-pipeline loop and of the bench line against
# the workgroups per region (GMX_COVER_SINGLE_BLOCKS; 8192 = the old grid of the queue's capacity at 1 M reads)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4 && out=gpurun_out/r4/cover_single_grid.txt && : > $out
for nb in 8192 1024 512 256 128; do
  echo "== GMX_COVER_SINGLE_BLOCKS=$nb" >> $out
  GMX_COVER_SINGLE_BLOCKS=$nb timeout 600 python tools/profile_config.py 3 1000000 8 2>&1 | grep -E "kernel pipeline|packed" >> $out
  GMX_COVER_SINGLE_BLOCKS=$nb timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('bench value', j['value'], 'kernel_pipeline', j.get('kernel_pipeline'), 'roofline', j['roofline']['achieved'])" >> $out
done
cat $out
