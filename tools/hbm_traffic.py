"""FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, kilobytes per dispatch) -> bytes per launch per kernel, as JSON.

FETCH_SIZE on gfx950 is TCC_EA0_RDREQ x 64 B (MI355X_MICROARCH.md, HBM section): exact for the 64-byte requests of
this engine's scattered accesses, half the true bytes for 128-byte streaming requests; `fetch_bytes_x2` is the
upper bound with the guide's doubling applied. Infinity-Cache hits are included (the counters sit at the L2's
fabric side), so this is an upper bound on HBM traffic.
"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("__amd") or "at::" in name:
                continue
            cell = acc[name][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
out = {}
for name, counters in sorted(acc.items()):
    f = counters.get("FETCH_SIZE", [0.0, 0])
    w = counters.get("WRITE_SIZE", [0.0, 0])
    fetch = f[0] / max(f[1], 1) * 1024.0
    write = w[0] / max(w[1], 1) * 1024.0
    out[name] = {"fetch_bytes": round(fetch), "fetch_bytes_x2": round(2 * fetch), "write_bytes": round(write),
                 "launches": max(f[1], w[1])}
print(json.dumps(out, indent=1))
