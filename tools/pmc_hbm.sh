#!/bin/bash
# HBM-side traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (never combined
# with a trace), summarised per kernel per launch into gpurun_out/hbm_traffic.json.
#   gpurun --timeout 900 -- 'bash tools/pmc_hbm.sh'
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/hbm_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv > gpurun_out/hbm_traffic.json
cat gpurun_out/hbm_traffic.json
