#!/bin/bash
# gpurun -- 'bash tools/ingest_timeline.sh OUT [n_reads] [quality model] [members per chunk]': start / end of every kernel and copy of the
# device-side ingestion's LAST pass over the file (tools/ingest_bench.py, INGEST_MAP=0), relative to the pass's first kernel
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/ingest_tl}; mkdir -p $OUT
N=${2:-4000000}; Q=${3:-binned}; STEP=${4:-8000}
INGEST_MAP=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o trace -- python tools/ingest_bench.py $N $Q $STEP > $OUT/prof.log 2>&1
python - <<PY | tee $OUT/timeline_${Q}_$STEP.txt
import csv, re, glob
ev = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob("$OUT/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Size", "")))
ev.sort()
infl = [i for i, e in enumerate(ev) if e[2].startswith("gmx_inflate")]
per_pass = 3 if len(infl) % 3 == 0 else 1
first = infl[-per_pass]
# the pass starts with the upload before its first inflate kernel
t0 = ev[first][0]
for s, e, n in ev:
    if s < t0 - 3_000_000: continue
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  {n}")
PY
rm -rf $OUT/t
