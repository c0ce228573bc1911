"""One configuration's roofline objects from its profile directory (tools/profile_round6.sh writes it):
  python tools/config_roofline.py N DIR > profiles/round6/configN_roofline.json
DIR holds kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/profile_config.py N), hbm_traffic.json (FETCH_SIZE / WRITE_SIZE
passes of the same command; tools/hbm_traffic.py) and loop_stats.txt (the same command with the -DGMX_LOOP_STATS build: iteration
mix of one batch). bench.py's config legs read the file: alg bytes and counter traffic from here, the kernels' durations live.

Byte model of gmx_extend_kernel (every term is something the kernel must move once per read whatever the cache
does): queue entry 4 + seed entry 8 + read planes 48 + TEXT steps x 32 (one 32-byte text record each) + HIT steps x 16 (a marker
sub-record) + path nodes 12 each (one per HIT step) + compact coverage record 32 + task id 4. TEXT and HIT steps per read are the
stats build's `heavy TEXT` / `heavy HIT` lane counts of one batch divided by its reads."""
import csv
import json
import re
import sys

which, d = sys.argv[1], sys.argv[2]
WORKLOAD = {"2": "configs[2]: 23.3 Mb + 2000 nested MSA regions (depth <= 3) + 100 k SNPs, k = 10",
            "3": "configs[3]: 64 444 167 bp + 1.8 M sites (90 % SNP / 10 % indel, 5 % multi-allelic), k = 14",
            "4": "configs[4]: 3.1 G bases + 85 M sites, k = 14"}
stats = {}
for row in csv.DictReader(open(f"{d}/kernel_stats.csv")):
    stats[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]), float(row["Percentage"]))
try:
    traffic = json.load(open(f"{d}/hbm_traffic.json"))
except (OSError, ValueError):
    traffic = {}
loop, sect, n_reads = {}, None, 1_000_000
try:
    for line in open(f"{d}/loop_stats.txt"):
        m0 = re.match(r"kernel pipeline: .* per (\d+) reads", line)
        if m0:
            n_reads = int(m0.group(1))
        if not line.startswith(" "):
            sect = line.split()[0] if line.split() else None
        elif sect in ("extend", "probe"):
            m = re.match(r"\s+(.*?)\s{2,}(\d+)\s+([\d.]+) per wave", line)
            if m:
                loop.setdefault(sect, {})[m.group(1).strip()] = float(m.group(2))
except OSError:
    pass


def short(name):  # "void gmx_extend_kernel<false, 1>(GmxIndexView, ...)" -> "gmx_extend_kernel<false, 1>" (the key tools/hbm_traffic.py uses)
    n = name[5:] if name.startswith("void ") else name
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i]
    return n


def find(prefix, exclude=None):
    for name in stats:
        if short(name).startswith(prefix) and not (exclude and exclude in name):
            return name
    return None


def tr(name):
    t = traffic.get(short(name)) if name else None
    return int(t["fetch_bytes"] + t["write_bytes"]) if t else None


out = {"config": WORKLOAD.get(which, which), "reads_per_launch": n_reads, "source": f"{d} (tools/profile_round6.sh)", "roofline": []}
ext = find("gmx_extend_kernel", ", 2>")
if ext and loop.get("extend"):  # (without the stats build's step counts there is no byte model for this configuration)
    both = {k: loop.get("extend", {}).get(k, 0) + loop.get("probe", {}).get(k, 0) for k in ("heavy TEXT", "heavy HIT", "lanes in heavy kinds", "lanes in slow iterations")}
    # lanes in heavy kinds = TEXT + HIT + WIDE lanes summed over iterations; split by the iterations' kinds
    e = loop.get("extend", {})
    lanes_heavy = e.get("lanes in heavy kinds", 0.0)
    it_text, it_hit, it_wide = e.get("heavy TEXT", 0.0), e.get("heavy HIT", 0.0), e.get("heavy WIDE", 0.0)
    tot_it = max(it_text + it_hit + it_wide, 1.0)
    text_steps = lanes_heavy * it_text / tot_it / n_reads if lanes_heavy else 3.4
    hit_steps = lanes_heavy * it_hit / tot_it / n_reads if lanes_heavy else 0.15
    slow_steps = e.get("lanes in slow iterations", 0.0) / n_reads
    alg = 4 + 8 + 48 + 32 * text_steps + (16 + 12) * hit_steps + 32 + 4 + 64 * slow_steps
    ns = stats[ext][0]
    out["roofline"].append({
        "bound": "hbm", "kernel": short(ext), "alg_bytes_per_read": round(alg, 1), "avg_launch_ms": ns / 1e6,
        "achieved": alg * n_reads / ns, "peak": 8000.0, "unit": "GB/s", "frac": alg * n_reads / ns / 8000.0, "traffic": tr(ext),
        "text_steps_per_read": round(text_steps, 2), "hit_steps_per_read": round(hit_steps, 2), "general_steps_per_read": round(slow_steps, 3),
        "alg_bytes_model": "queue entry 4 + seed entry 8 + read planes 48 + TEXT steps x 32 + HIT steps x (16 + 12) + general iterations x 64 + coverage record 32 + task id 4; steps from the stats build (loop_stats.txt: lanes served by heavy iterations, split over TEXT and HIT in proportion to the iterations that ran each kind — the build counts lanes per iteration, not per kind)"})
for prefix, per_task, model in (("gmx_probe_kernel", 217, "per (read, orientation), with the screening side table (round 6): k-mer table entry 8 + read planes 48 + the entry's count word 4 + 16 side words 64 + 1.3 candidates x (header 24 + text record 32) + queue / parked entries ~20 (460 with the header walk of rounds 4-5: 16 states x ~24 B)"),
                                ("gmx_seed_kernel", 24 + 6, "per task: the read's last plane pair 16 + seed directory entry 8; + 12 per alive task")):
    k = find(prefix)
    if k:
        ns = stats[k][0]
        b = per_task * 2 * n_reads
        out["roofline"].append({"bound": "hbm", "kernel": short(k), "alg_bytes_per_task": per_task, "tasks_per_launch": 2 * n_reads, "avg_launch_ms": ns / 1e6,
                                "achieved": b / ns, "peak": 8000.0, "unit": "GB/s", "frac": b / ns / 8000.0, "traffic": tr(k), "alg_bytes_model": model})
for prefix in ("gmx_cover_jump_kernel", "gmx_cover_single_kernel"):
    k = find(prefix)
    if k:
        ns = stats[k][0]
        sites = {"2": 1.0, "3": 4.0, "4": 4.0}.get(which, 2.0)
        alg = 32 + 4 + sites * (32 + 8)
        out["roofline"].append({"bound": "hbm", "kernel": short(k), "alg_bytes_per_read": alg, "avg_launch_ms": ns / 1e6, "achieved": alg * n_reads / ns, "peak": 8000.0,
                                "unit": "GB/s", "frac": alg * n_reads / ns / 8000.0, "traffic": tr(k),
                                "alg_bytes_model": f"compact record 32 + task id 4 + {sites:g} sites crossed x (a 32 B site record + one 8 B atomic) per mapped read"})
out["kernel_time_shares"] = {short(n): {"avg_ms": round(v[0] / 1e6, 4), "calls": v[1], "percent": v[2]} for n, v in sorted(stats.items(), key=lambda kv: -kv[1][2])[:8]}
print(json.dumps(out, indent=1))
