#!/bin/bash
# SQ counters of gmx_inflate_kernel: gpurun -- 'bash tools/exp/ingest_pmc.sh OUT'
set -u
export TMPDIR=/tmp INGEST_MAP=0
OUT=${1:-gpurun_out/ingest_pmc}; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python tools/ingest_bench.py 1000000 binned 7168 > $OUT/$name.log 2>&1; echo "pmc $name rc=$?"; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH
run c SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_CYCLES
python tools/pmc_summary.py $OUT/a/pmc_counter_collection.csv $OUT/b/pmc_counter_collection.csv $OUT/c/pmc_counter_collection.csv 2>/dev/null | grep -A30 "inflate" | head -40 | tee $OUT/summary.txt
rm -rf $OUT/a $OUT/b $OUT/c
