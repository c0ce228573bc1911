#!/bin/bash
# gpurun -- 'bash tools/ingest_prof.sh OUT [n_reads] [quality model] [members per chunk]': rates + kernel stats of the device-side ingestion
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/ingest}; mkdir -p $OUT
N=${2:-4000000}; Q=${3:-binned}; STEP=${4:-8000}
python tools/ingest_bench.py $N $Q $STEP 2>&1 | grep -v amdgpu.ids | tee $OUT/rates_${Q}_$STEP.txt
INGEST_MAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o trace -- python tools/ingest_bench.py $N $Q $STEP > $OUT/prof.log 2>&1
python - <<PY | tee $OUT/kernel_stats_${Q}_$STEP.txt
import csv, re
for i, r in enumerate(csv.reader(open("$OUT/t/trace_kernel_stats.csv"))):
    if i == 0 or i > 12: continue
    name = re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0]
    print(f"{name:28s} calls {r[1]:>4s}  avg {float(r[3]) / 1e3:10.1f} us  total {float(r[2]) / 1e6:8.2f} ms  {r[4]:>6s} %")
PY
rm -rf $OUT/t
