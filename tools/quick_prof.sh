#!/bin/bash
# bench value twice + per-kernel stats + HBM-side traffic of the search kernels: gpurun -- 'bash tools/quick_prof.sh OUT'
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/quick}; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],4), 'ms/step')" | tee -a $OUT/values.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench_prof.log 2>&1
cut -c1-110 $OUT/trace/trace_kernel_stats.csv | head -8 | tee $OUT/kernel_stats_head.txt
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv > $OUT/step_timeline.txt 2>&1; rm -f $OUT/trace/trace_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
done
python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
rm -rf $OUT/pmc_*/pmc_counter_collection.csv
python - <<PY
import json
d=json.load(open("$OUT/hbm_traffic.json"))
for k in d:
    if any(x in k for x in ("extend","seed_kernel","cover_single","pack")): print(k[:40], d[k])
PY
