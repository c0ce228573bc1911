#!/bin/bash
# One step of the bench loop as a kernel timeline (start, end, duration in us; queue).  gpurun -- 'bash tools/bench_timeline.sh OUT'
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/bt}; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench.txt 2>&1
python - <<PY
import csv
rows = sorted(csv.DictReader(open("$OUT/trace/trace_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "pack" in r["Kernel_Name"]]
i0, i1 = packs[6], packs[8]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:70]}")
PY
rm -f $OUT/trace/trace_kernel_trace.csv
