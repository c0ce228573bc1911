#!/bin/bash
# One SQ counter pass over the bench (instruction mix per kernel).  gpurun --timeout 600 -- 'bash tools/pmc_sq.sh'
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sq1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq1 -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq1.log 2>&1
echo "rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_sq1/pmc_counter_collection.csv > gpurun_out/pmc_sq_summary.txt
grep -A8 "probe\|extend\|cover_kernel<CoverEnvT<32u, 8u, 64u, 64u>, 0>" gpurun_out/pmc_sq_summary.txt
