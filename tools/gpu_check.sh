#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, and a kernel-trace profile of the bench.
# Usage (from the repo root):  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench1.log
rm -rf gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?"; find gpurun_out/prof_r1 -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-160 | head -14
