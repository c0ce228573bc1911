#!/bin/bash
# Kernels AND host-to-device copies of the packed host feed on one time axis (rocprofv3 --kernel-trace --memory-copy-trace):
# does the upload of batch i+1 run beside the kernels of batch i?  Usage: tools/feed_timeline.sh [CONFIG=3] [OUT=gpurun_out/feedtl]
set -u
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
C=${1:-3}; OUT=${2:-gpurun_out/feedtl}; mkdir -p $OUT; rm -rf /tmp/feedtl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/feedtl -o t -- python tools/profile_config.py $C 1000000 8 > $OUT/run_c$C.txt 2>&1
grep -E "kernel pipeline|packed host feed" $OUT/run_c$C.txt
python - "$C" "$OUT" <<'PY'
import csv, glob, sys
c, out = sys.argv[1], sys.argv[2]
k = glob.glob("/tmp/feedtl/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("/tmp/feedtl/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:60])))
for r in csv.DictReader(open(m)):
    name = r.get("Direction") or r.get("Name") or "copy"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (name, r.get("Size", r.get("Bytes", "?")))))
ev.sort()
# the last three batches of the run (the packed feed leg comes last): from the third-last gmx_batch_begin_kernel on
begins = [i for i, e in enumerate(ev) if "gmx_batch_begin_kernel" in e[2]]
i0 = begins[-4]
t0 = ev[i0][0]
with open(f"{out}/timeline_c{c}.txt", "w") as fh:
    for s, e, what in ev[i0:]:
        line = f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {what}"
        fh.write(line + "\n")
        if "COPY" in what and (e - s) > 50_000 or "batch_begin" in what or "extend_kernel" in what or "cover_jump" in what:
            print(line)
PY
rm -rf /tmp/feedtl
