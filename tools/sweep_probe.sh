#!/bin/bash
# bench at several probe budgets.  gpurun --timeout 900 -- 'bash tools/sweep_probe.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for it in 0 4 6 8 10 14 20; do
  echo "GMX_PROBE_ITERS=$it"
  GMX_PROBE_ITERS=$it python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernels_ms_per_launch'])"
done
