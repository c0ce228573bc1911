#!/bin/bash
# A/B of two builds of libgmx.so on ONE GPU box (boxes differ by several per cent, so numbers from different gpurun
# calls do not compare). Put the baseline build at gramtools_amd/lib/libgmx_prev.so (e.g. `git archive <rev>
# gramtools_amd/csrc include | tar -x -C /tmp/prev` and hipcc with the flags of gramtools_amd/build.py), then
#   gpurun --timeout 1200 -- 'bash tools/ab_bench.sh [ROUNDS]'
export TMPDIR=/tmp
L=gramtools_amd/lib
cp $L/libgmx.so $L/libgmx_new.so
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6), 'M reads/s', round(d['ms_per_step'], 4), 'ms/step')"; }
for rep in $(seq 1 ${1:-3}); do
  cp $L/libgmx_prev.so $L/libgmx.so; run prev
  cp $L/libgmx_new.so $L/libgmx.so; run new
done
