#!/bin/bash
# SQ counters of gmx_inflate_kernel (instruction mix, scalar unit) over one pass of tools/ingest_bench.py:
#   gpurun --timeout 600 -- 'bash tools/pmc_ingest_sq.sh' -> gpurun_out/ingest_inflate_sq_counters.txt
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_ing1 gpurun_out/pmc_ing2
INGEST_MAP=0 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ing1 -o pmc -- python tools/ingest_bench.py 1600000 binned 8000 > gpurun_out/pmc_ing1.log 2>&1
echo "rc=$?"
INGEST_MAP=0 timeout 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_ing2 -o pmc -- python tools/ingest_bench.py 1600000 binned 8000 > gpurun_out/pmc_ing2.log 2>&1
echo "rc=$?"
python tools/pmc_summary.py $(find gpurun_out/pmc_ing1 gpurun_out/pmc_ing2 -name "*counter_collection.csv") | grep -A20 "gmx_inflate_kernel" > gpurun_out/ingest_inflate_sq_counters.txt
cat gpurun_out/ingest_inflate_sq_counters.txt
rm -rf gpurun_out/pmc_ing1 gpurun_out/pmc_ing2
