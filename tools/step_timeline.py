"""One bench step as a timeline, from a rocprofv3 kernel trace (tools/gpu_check.sh writes gpurun_out/prof_r1).
Usage: python tools/step_timeline.py [TRACE.csv]"""
import csv
import glob
import sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof_r1/**/*kernel_trace.csv", recursive=True))[0]
first = sys.argv[2] if len(sys.argv) > 2 else "pack"   # the batch's first kernel: gmx_pack_kernel (bytes) or gmx_batch_begin_kernel (bit planes)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0, i1 = packs[-3], packs[-2]
t0 = int(rows[i0]["Start_Timestamp"])
print("start_us   end_us   dur_us  queue kernel")
for r in rows[i0:i1 + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:72]}")
