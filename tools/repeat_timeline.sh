#!/bin/bash
# configs[1] with a fraction of the genome in 10-copy repeats: device-resident rate and one step's kernel timeline.
#   gpurun -- 'bash tools/repeat_timeline.sh 0.002 OUT'
set -u
export TMPDIR=/tmp
FRAC=${1:-0.002}; OUT=${2:-gpurun_out/rep}; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$FRAC -o trace -- python tools/scale_check.py 4411532 60000 10 1000000 repeats=$FRAC > $OUT/scale_$FRAC.txt 2>&1
grep "device-resident\|queues of the last" $OUT/scale_$FRAC.txt | cut -c1-400
python - <<PY
import csv
rows = sorted(csv.DictReader(open("$OUT/trace_$FRAC/trace_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "pack" in r["Kernel_Name"]]
i0, i1 = packs[-2], packs[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:80]}")
PY
rm -f $OUT/trace_$FRAC/trace_kernel_trace.csv
