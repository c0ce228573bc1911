#!/bin/bash
# bench under several environment settings.  gpurun --timeout 900 -- 'bash tools/sweep_env.sh VAR v1 v2 ...'
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],3), 'ms; extend', round(d['roofline']['avg_launch_ms'],3), 'other', round(d['roofline']['other_kernels_ms_per_launch'],3))"
done
