"""gmx_extend_kernel's issue-side figures from one profile directory (tools/profile_round2.sh): VALU busy share of the
SIMDs, active-lane share of the wave loop, iterations per wave — the `roofline.issue` object of bench.py.
  valu_busy         = SQ_INSTS_VALU x 4 cycles (a wave64 VALU instruction occupies its SIMD16 for 4 cycles)
                      / (kernel duration x SIMD clock x number of SIMDs)
  active_lane_share = lanes holding a search state, summed over the loop's iterations / (iterations x 64 lanes)"""
import csv
import json
import re
import sys

d = sys.argv[1]
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; MI355X peak engine clock
sq = {}
name = None
for line in open(f"{d}/sq_counters.txt"):
    if not line.startswith(" "):
        name = line.strip()
    elif name and "gmx_extend_kernel" in name and ", 2>" not in name:  # (the first pass; ", 2>" is the stragglers' pass)
        k, v = line.split()[:2]
        sq[k] = float(v)
dur_ns = None
for row in csv.DictReader(open(f"{d}/bench_kernel_stats.csv")):
    if "gmx_extend_kernel" in row["Name"] and ", 2>" not in row["Name"]:
        dur_ns = float(row["AverageNs"])
loop = {}
sect = None
for line in open(f"{d}/loop_stats.txt"):
    if not line.startswith(" "):
        sect = line.split()[0]
    elif sect == "extend":
        m = re.match(r"\s+(.*?)\s{2,}(\d+)\s+([\d.]+) per wave", line)
        if m:
            loop[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)))
iters = loop["fast iterations"][1] + loop["slow iterations"][1]
live = loop.get("live lanes (sum over iterations)", (0.0, 0.0))[1]
out = {
    "kernel": "gmx_extend_kernel", "avg_duration_ns": dur_ns,
    "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"), "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
    "vmem_rd_insts_per_launch": sq.get("SQ_INSTS_VMEM_RD"), "vmem_wr_insts_per_launch": sq.get("SQ_INSTS_VMEM_WR"),
    "waves_launched": sq.get("SQ_WAVES"), "waves_with_work": loop["waves"][0],
    "valu_busy": sq["SQ_INSTS_VALU"] * 4 / (dur_ns * 1e-9 * CLOCK_HZ * N_SIMD),
    "wait_share_of_wave_cycles": sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
    "iterations_per_wave": iters, "heavy_steps_per_lane": loop["lanes in heavy kinds"][1] / 64,
    "active_lane_share": live / (loop["fast iterations"][1] * 64),
}
print(json.dumps(out, indent=1))
