#!/bin/bash
# Hardware-counter passes over the bench (one rocprofv3 run per counter set; --pmc is never combined with a trace).
# Usage (repo root): gpurun --timeout 1500 -- 'bash tools/pmc.sh'
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCC_ATOMIC_WITH_RET_REQ TCP_TCC_ATOMIC_WITHOUT_RET_REQ
run tcc TCC_HIT TCC_MISS TCC_REQ TCC_ATOMIC
python tools/pmc_summary.py gpurun_out/pmc_*/pmc_counter_collection.csv > gpurun_out/pmc_summary.txt
cat gpurun_out/pmc_summary.txt
