#!/bin/bash
# The round-5 profiles (copy what is to be judged into profiles/round5/):  gpurun --timeout 2400 -- 'bash tools/profile_round5.sh'
#   1. GPU test suite                                          -> pytest_gpu.log
#   2. bench.py as the driver runs it                          -> bench.log (the JSON line)
#   3. rocprofv3 --kernel-trace --stats of the same command    -> bench_kernel_stats.csv, step timelines
#   4. FETCH_SIZE / WRITE_SIZE passes                          -> hbm_traffic.json
#   5. device-side BGZF ingestion: rates, kernel stats, CLI    -> ingest_*.txt, cli_bgzf.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r5
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
python tools/step_timeline.py $OUT/trace/trace_kernel_trace.csv gmx_batch_begin > $OUT/step_timeline_host_feed.txt 2>&1
rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/hbm_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/hbm_traffic.json
rm -rf $OUT/pmc_*
timeout 600 bash tools/ingest_prof.sh $OUT 4000000 binned 7168 > $OUT/ingest_binned.log 2>&1
timeout 300 env INGEST_MAP=0 python tools/ingest_bench.py 4000000 wide 7168 2>&1 | grep -v amdgpu > $OUT/ingest_rates_wide.txt
timeout 600 python tools/cli_bgzf.py 4000000 2>&1 | grep -v amdgpu > $OUT/cli_bgzf.txt
tail -1 $OUT/bench.log | cut -c1-400
