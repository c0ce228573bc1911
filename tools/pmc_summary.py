"""Per-kernel mean of every collected counter per dispatch (rocprofv3 --pmc csv outputs)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if name.startswith("__amd") or "at::" in name or "nccl" in name.lower():
                continue
            cell = acc[name][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
for name in sorted(acc):
    print(name)
    for counter in sorted(acc[name]):
        total, n = acc[name][counter]
        print(f"    {counter:36s} {total / n:16.1f}   (mean of {n} dispatches)")
