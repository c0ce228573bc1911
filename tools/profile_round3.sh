#!/bin/bash
# Every profile DESIGN.md / bench.py cite for round 3, written under gpurun_out/prof_r3 (copy what is to be judged into
# profiles/round3/):  gpurun --timeout 3000 -- 'bash tools/profile_round3.sh'
#   1. bench.py as the driver runs it                                    -> bench.log
#   2. rocprofv3 --kernel-trace --stats of the same command              -> bench_kernel_stats.csv, step_timeline_*.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)              -> hbm_traffic.json
#   4. rocprofv3 --pmc SQ counters (own passes)                          -> sq_counters.txt, sq_extend.json (+ loop statistics)
#   5. configs[2] / configs[3] by their full recipes: kernel stats, rates, queues; SQ + HBM counters of configs[2]
#   6. configs[1] with 0.2 % / 5 % of the genome in 10-copy repeats      -> repeat_timeline_*.txt
#   7. the executable's feed: events of one run, the parser alone        -> cli_trace.txt, parse_bench.txt
#   8. index build phases at chr20 scale and for configs[2]              -> build_trace_*.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
STATS=$PWD/gramtools_amd/lib/libgmx_stats.so
python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv pack > $OUT/step_timeline_device_resident.txt 2>&1
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv gmx_batch_begin > $OUT/step_timeline_host_feed.txt 2>&1
rm -rf $OUT/trace/trace_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
python tools/pmc_summary.py $OUT/pmc_sq1/pmc_counter_collection.csv $OUT/pmc_sq2/pmc_counter_collection.csv > $OUT/sq_counters.txt
rm -rf $OUT/pmc_*/pmc_counter_collection.csv $OUT/pmc_*/*agent_info.csv
GMX_LIB=$STATS python tools/loop_stats.py > $OUT/loop_stats.txt 2> $OUT/loop_stats.err
python tools/sq_extend.py $OUT > $OUT/sq_extend.json 2> $OUT/sq_extend.err
# ---- configs[2], configs[3] ----
for C in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$C -o trace -- python tools/scale_check_configs.py $C 1000000 > $OUT/config${C}_run.txt 2>&1
  cp $OUT/c$C/trace_kernel_stats.csv $OUT/config${C}_kernel_stats.csv 2>/dev/null
  python tools/step_timeline.py $OUT/c$C/trace_kernel_trace.csv pack > $OUT/config${C}_step_timeline.txt 2>&1
  rm -rf $OUT/c$C
  grep -E "configs|index:|device-resident|queues" $OUT/config${C}_run.txt | cut -c1-500 > $OUT/config${C}_rate.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/c2pmc_$c -o pmc -- python tools/scale_check_configs.py 2 1000000 > /dev/null 2>&1
done
python tools/hbm_traffic.py $OUT/c2pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/c2pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/config2_hbm_traffic.json
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/c2sq1 -o pmc -- python tools/scale_check_configs.py 2 1000000 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/c2sq2 -o pmc -- python tools/scale_check_configs.py 2 1000000 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/c2sq1/pmc_counter_collection.csv $OUT/c2sq2/pmc_counter_collection.csv > $OUT/config2_sq_counters.txt
rm -rf $OUT/c2pmc_* $OUT/c2sq1 $OUT/c2sq2
GMX_LIB=$STATS python tools/coop_stats_c2.py > $OUT/coop_phases_config2.txt 2>&1
# ---- repeats ----
for F in 0.002 0.05; do bash tools/repeat_timeline.sh $F $OUT/rep 2>&1 | grep -v "^[EW]2026" > $OUT/repeat_timeline_$F.txt; done
rm -rf $OUT/rep
# ---- feed ----
tools/cli_trace.sh 4000000 64 > $OUT/cli_trace.txt 2>&1
python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from bench import write_fastq
write_fastq("/tmp/r4m.fq", [np.random.default_rng(1).integers(1, 5, size=(4000000, 150), dtype=np.uint8)])
PY
for t in 16 32 64; do echo "threads $t"; gramtools_amd/bin/gram _parse_bench /tmp/r4m.fq $t 3; done > $OUT/parse_bench.txt 2>&1
echo "pread path (GMX_FASTQ_MMAP=0), 64 threads" >> $OUT/parse_bench.txt; GMX_FASTQ_MMAP=0 gramtools_amd/bin/gram _parse_bench /tmp/r4m.fq 64 3 >> $OUT/parse_bench.txt 2>&1
rm -f /tmp/r4m.fq
# ---- index build ----
GMX_BUILD_TRACE=1 python tools/build_trace.py 64444167 1800000 14 > $OUT/build_trace_chr20_snps.txt 2>&1
GMX_DEVICE_BUILD=0 GMX_BUILD_TRACE=1 python tools/build_trace.py 64444167 1800000 14 > $OUT/build_trace_chr20_snps_host_walk.txt 2>&1
tail -3 $OUT/bench.log | cut -c1-600
