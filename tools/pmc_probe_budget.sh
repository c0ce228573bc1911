#!/bin/bash
# Instruction counts of the probe kernel at two iteration budgets: the difference is the cost of the wave loop.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for it in 1 10; do
  rm -rf gpurun_out/pmc_pb$it
  GMX_PROBE_ITERS=$it timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_pb$it -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_pb$it.log 2>&1
  echo "== GMX_PROBE_ITERS=$it"
  python tools/pmc_summary.py gpurun_out/pmc_pb$it/pmc_counter_collection.csv | grep -A6 "^gmx_probe\|^gmx_extend"
done
