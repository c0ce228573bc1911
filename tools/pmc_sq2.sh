#!/bin/bash
# Two SQ counter passes over the bench (instruction mix + wait breakdown per kernel).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local name=$1; shift; rm -rf gpurun_out/pmc_$name; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
run sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
python tools/pmc_summary.py gpurun_out/pmc_sq1/pmc_counter_collection.csv gpurun_out/pmc_sq2/pmc_counter_collection.csv gpurun_out/pmc_sq3/pmc_counter_collection.csv > gpurun_out/pmc_sq_summary.txt
grep -A25 "^gmx_seed\|^gmx_extend\|^gmx_cover_single" gpurun_out/pmc_sq_summary.txt
