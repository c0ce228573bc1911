#!/usr/bin/env python
"""bench.py — 150 bp reads quasimapped per second (BASELINE.json metric), whole job over N GPUs.

Workload (config.workload): BASELINE.json configs[1] — M. tuberculosis scale: 4 411 532 bp random reference +
60 000 SNP sites written as a PRG, k = 10, 1 M x 150 bp error-free reads per GPU and step (50 % reverse strand),
synthetic (no real genomes offline).

TIMED REGION (since round 3) = SURVEY.md §8(d)'s: host buffers in -> coverage arrays final on the host. A job is
  zeroed accumulators -> K steps -> (N > 1) ONE exchange of the coverage (RCCL all-reduce of the fused block, driven from
  inside the library: gmx_comm_allreduce_coverage, the routine `gram genotype --devices` uses) -> D2H of the accumulator
  block + gather into the three coverage arrays (gmx_coverage_fetch).
A step = one batch of DISTINCT reads (8 batches are cycled: 300 MB of packed reads, more than the 256 MiB Infinity Cache)
handed over as a 2-bit stream in page-locked host memory (gmx_map_reads_2bit_host: 37.5 B per 150 bp read, unpacked to bit
planes by the batch's first kernel; --planes: bit planes, gmx_map_reads_packed_host, 40 B per read, the form the FASTQ parser
threads of `gram` emit; SURVEY.md §7 "pre-encoded (2-bit packed ...) pinned-host" input), the per-read seeds left in page-locked
host memory where the few reads that draw read theirs in place (--upload-seeds: + 4 B per read): H2D on a copy stream beside
the kernels of the batches before, then the whole kernel pipeline (seed look-up, search, k-mer filter, selection, coverage
atomics; forward and reverse complement). FASTQ parsing and the index build are outside (reported separately).
Rounds 1 and 2 timed a narrower region as `value` (reads resident in HBM, one batch replayed, coverage left in HBM): that
figure is still on the line as `kernel_pipeline` — the two are NOT comparable.

Extra keys (rank 0; the side legs run at N = 1 only unless noted):
  kernel_pipeline  reads resident in HBM (one byte per base), the same batch every step, coverage left in HBM
  sustained        the `value` loop for >= 1 s
  exchange_ms      (N > 1) the coverage exchange alone, timed between fences after the job
  per_rank_s       (N > 1) every rank's own time for the timed job
  cli_end_to_end   the `gram` executable on a FASTQ file: parse + upload + map + the three coverage files
  bgzf_device_feed the reads as a BGZF FASTQ decoded by HIP kernels (inflate, CRC, record scan, packing) and mapped
  cpu_baseline     the oracle (CPU restatement of the reference algorithm, "port"): the cores the container grants, and one thread
  roofline         what bounds the STEP (the host link: bound "pcie", achieved = H2D GB/s of the timed job against the
                   PCIe Gen5 x16 spec) and, in roofline.kernels, one object per leading kernel — gmx_extend_kernel first (the
                   dominant one: HBM fraction by the byte model of DESIGN.md §8, by counter traffic, and its issue fraction),
                   then the single-instance coverage kernel, the k-mer filter and the seed look-up — each with the duration
                   of its own dispatch measured live (HIP events attached to the dispatch inside the library)
  configs          (N = 1) BASELINE.json configs[2], configs[3] and (on a box with >= 280 GiB of host memory) configs[4] built at
                   full size: kernel-pipeline and packed-feed reads/s, their leading kernels' durations and roofline objects
  jobs             the timed job (exactly --steps steps) is run as often as it takes to add up to 1 s; `value` is the MEDIAN job
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME = 4411532
N_SITES = 60000
KMER = 10
READ_LEN = 150
READS_PER_GPU = 1_000_000
N_BATCHES = 8                                              # distinct batches cycled by the timed loop
B_NOMINAL_PER_READ = 128 * (READ_LEN - KMER) + READ_LEN   # 18 070 B/read at k = 10
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0                                       # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec), per direction
# Algorithmic bytes gmx_extend_kernel must move per mapped read with text-form states (HISTORY.md §8 derives each term):
# queue entry 4 + seed directory entry 8 + packed read planes 48 + PRG text records 6 x 16 + marker sub-records 3 x 16 +
# path nodes 2 x 12 + coverage record 32 + task id 4
# what gmx_extend_kernel has to move per mapped read since round 3 (HISTORY.md §8): queue entry 4 + seed directory entry 8 + read
# planes 48 + PRG text records 3.4 x 32 (64 symbols each; loop_stats.txt: 3.3 heavy steps per lane) + marker sub-records of the
# sites that are not inline 0.15 x 16 + path nodes 2 x 12 + coverage record 32 + task id 4   (264 B with round 2's 16 B records)
B_DESIGN_PER_READ = 4 + 8 + 48 + 109 + 2 + 2 * 12 + 32 + 4
PROFILE_DIRS = [os.path.join(ROOT, "profiles", "round6"), os.path.join(ROOT, "profiles", "round5"), os.path.join(ROOT, "profiles", "round4"), os.path.join(ROOT, "profiles", "round3"), os.path.join(ROOT, "profiles", "round2")]


def profile_json(name):
    for d in PROFILE_DIRS:
        try:
            with open(os.path.join(d, name)) as fh:
                return json.load(fh), os.path.relpath(os.path.join(d, name), ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (FETCH_SIZE + WRITE_SIZE, separate --pmc passes; tools/pmc_hbm.sh)."""
    table, src = profile_json("hbm_traffic.json")
    try:
        k = next(v for name, v in table.items() if name.startswith(kernel))  # template instances: gmx_extend_kernel<...>
        return int(k["fetch_bytes"] + k["write_bytes"]), src
    except (TypeError, KeyError, AttributeError, StopIteration):
        return None, src


def effective_cores():
    """Cores' worth of CPU time this process can use: the hardware threads, or the container's quota if that is lower (the
    GPU boxes show 256 hardware threads and grant 16 cores: cgroup cpu.max = 1600000 100000; tools/exp/cpu_probe.cpp)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(prg, reads, seeds, max_seconds=12.0):
    """Oracle (port of the reference algorithm, OpenMP over reads as quasimap.cpp:90) on a bounded sample of the same reads."""
    from oracle import Oracle
    from gramtools_amd.synth import flat_offsets
    cores = effective_cores()
    o = Oracle(prg, KMER)
    n1 = 3000
    t0 = time.time()
    o.map_reads(reads[:n1].reshape(-1), flat_offsets(n1, READ_LEN), seeds[:n1], threads=1)
    one = n1 / max(time.time() - t0, 1e-6)
    n1 = int(min(reads.shape[0], max(n1, one * 5.0)))
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n1].reshape(-1), flat_offsets(n1, READ_LEN), seeds[:n1], threads=1)
    dt1 = time.time() - t0
    n = 4000
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n].reshape(-1), flat_offsets(n, READ_LEN), seeds[:n], threads=cores)
    rate = n / max(time.time() - t0, 1e-6)
    n2 = int(min(reads.shape[0], 400_000, max(n, rate * max_seconds * 0.6)))  # OpenMP scaling is sub-linear: keep it bounded
    o.reset_coverage()
    t0 = time.time()
    o.map_reads(reads[:n2].reshape(-1), flat_offsets(n2, READ_LEN), seeds[:n2], threads=cores)
    dt = time.time() - t0
    return {"value": n2 / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {n2} of the rank-0 reads of batch 0, same PRG/k/seeds, OpenMP over reads ({cores} threads = the cores' "
                      f"worth of CPU time the container grants, of {os.cpu_count()} hardware threads), {dt:.1f} s",
            "single_thread": {"value": n1 / dt1, "unit": "reads/s", "cores": 1, "sample": f"first {n1} reads, {dt1:.1f} s"}}


def write_fastq(path, batches):
    """Four-line FASTQ of uint8 reads (1..4), fixed-width names, quality 'I': numpy only."""
    name_w = 10
    first = 0
    with open(path, "wb") as fh:
        for reads in batches:
            n, L = reads.shape
            row = np.empty((n, 1 + name_w + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
            row[:, 0] = ord("@")
            idx = np.arange(first, first + n)
            for d in range(name_w):
                row[:, 1 + d] = (idx // 10 ** (name_w - 1 - d)) % 10 + ord("0")
            row[:, 1 + name_w] = ord("\n")
            row[:, 2 + name_w:2 + name_w + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[reads - 1]
            o = 2 + name_w + L
            row[:, o] = ord("\n")
            row[:, o + 1] = ord("+")
            row[:, o + 2] = ord("\n")
            row[:, o + 3:o + 3 + L] = ord("I")
            row[:, o + 3 + L] = ord("\n")
            row.tofile(fh)
            first += n


def _bgzf_piece(args):
    """(worker) records of `reads` (uint8 1..4) with Illumina-style headers and binned qualities, as BGZF members."""
    import struct
    import zlib
    seed, first, reads = args
    rng = np.random.default_rng(seed)
    n, L = reads.shape
    bases = np.frombuffer(b"NACGT", dtype=np.uint8)[reads]
    lvl = np.frombuffer(b"F:,#", dtype=np.uint8)
    q = lvl[np.repeat(rng.choice(4, size=(n, (L + 4) // 5), p=[0.9, 0.06, 0.03, 0.01]), 5, axis=1)[:, :L]]
    xs, ys = rng.integers(1000, 30000, n), rng.integers(1000, 30000, n)
    text = b"".join(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGTACGT\n%s\n+\n%s\n" % (1101 + (first + i) // 40000, xs[i], ys[i], bases[i].tobytes(), q[i].tobytes())
                    for i in range(n))
    out = bytearray()
    for i in range(0, len(text), 65280):
        p = text[i:i + 65280]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(p) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p))
    return bytes(out), len(text)


def bgzf_device_feed(ix, reads, seeds_np):
    """SURVEY 8f-3 on the device (gmx_ingest_*): the batch's reads as a BGZF FASTQ (bgzip's members, zlib level 6) in page-locked
    memory -> upload of the compressed members, inflate + CRC, record scan, 2-bit packing, quasimap — nothing inflated or
    parsed on the host. Rate of the whole chain, and of the decoding alone."""
    from concurrent.futures import ThreadPoolExecutor  # (threads: zlib releases the GIL; a fork of a process that holds a HIP context is not safe)
    from gramtools_amd import Ingest, PinnedArray, Quasimapper, bgzf_members
    n = reads.shape[0]
    per = 25000
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 8)) as ex:
        parts = list(ex.map(_bgzf_piece, [(11 + i, i * per, reads[i * per:(i + 1) * per]) for i in range((n + per - 1) // per)]))
    data = b"".join(p for p, _ in parts)
    text_bytes = sum(t for _, t in parts)
    mem = bgzf_members(data)
    pin = PinnedArray(len(data) + 64, np.uint8)
    pin.array[:len(data)] = np.frombuffer(data, dtype=np.uint8)
    step = 7168
    chunks = [mem[i:i + step] for i in range(0, len(mem), step)]
    ing = Ingest(max_text_bytes=min(len(mem), step) * 65536 + (1 << 20))
    # (the member tables as the C ABI takes them, built ahead: `gram` walks the file's member table on a thread of its own)
    arrays = [Ingest.member_array([(o - ch[0][0], s_, i_, c_) for o, s_, i_, c_ in ch]) for ch in chunks]
    seeds = PinnedArray(n, np.uint32)
    seeds.array[:] = seeds_np[:n]
    qm = Quasimapper(ix)

    def run(mapped):
        ing.reset()
        if mapped:
            qm.reset()
        total, at = 0, 0

        def submit(ci):
            ch = chunks[ci]
            lo, hi = ch[0][0], ch[-1][0] + ch[-1][1]
            ing.submit_bgzf(ci % 3, pin.array[lo:hi], arrays[ci], ci == len(chunks) - 1)
        t0 = time.perf_counter()
        submit(0)
        if len(chunks) > 1:
            submit(1)
        for ci in range(len(chunks)):
            if ci + 2 < len(chunks):  # (three slots: chunk ci + 2 takes chunk ci - 1's, waited for and released)
                submit(ci + 2)
            res = ing.wait(ci % 3)
            if res.status:
                raise RuntimeError(f"gmx_ingest status {res.status} at member {res.bad_member}")
            if mapped:
                qm.map_ingested(res, seeds, first=at)
                ing.release_after(ci % 3, engine=qm)
            at += int(res.n_reads)
            total += int(res.n_reads)
        if mapped:
            qm.sync()
        return total, time.perf_counter() - t0
    run(True)
    dec = sorted(run(False)[1] for _ in range(3))[1]
    both = sorted(run(True)[1] for _ in range(3))[1]
    st = qm.coverage().stats.as_dict()
    ing.close()
    out = {"reads": n, "bgzf_bytes_per_read": len(data) / n, "text_bytes_per_read": text_bytes / n, "members": len(mem),
           "decode_only": {"value": n / dec, "unit": "reads/s", "text_GBps": text_bytes / dec / 1e9},
           "decode_and_quasimap": {"value": n / both, "unit": "reads/s"}, "exact_mapped": st["exact_mapped"],
           "bound": "gmx_inflate_kernel: the CU's scalar unit (one thread of control per wavefront, 339 k scalar instructions per 64 KB member; profiles/round5/ingest_inflate_sq_counters.txt), the chunks' inflate kernels side by side (three slots, streams of their own hardware queues)",
           "host_feed_for_comparison": "BGZF inflated by 16 host cores: 32-48 M reads/s (profiles/round4/gz_feed.txt)"}
    pin.close()
    seeds.close()
    return out


def cli_end_to_end(prg, batches, threads):
    """`gram build` + `gram genotype` on a FASTQ of these reads: what the Python front-end's subprocess call costs."""
    from gramtools_amd.build import build_gram
    gram = build_gram()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        np.asarray(prg, dtype="<u4").tofile(os.path.join(d, "prg"))
        fq = os.path.join(d, "reads.fastq")
        write_fastq(fq, batches)
        t0 = time.time()
        b = subprocess.run([gram, "build", "--gram_dir", d, "--kmer_size", str(KMER), "--max_threads", str(threads)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_build = time.time() - t0
        if b.returncode:
            return {"error": b.stdout[-400:]}
        def one_call(n_threads, tag, extra_env=None):
            env = dict(os.environ)
            env.update(extra_env or {})
            t0 = time.time()
            g = subprocess.run([gram, "genotype", "--gram_dir", d, "--reads", fq, "--sample_id", "bench", "--ploidy", "haploid",
                                "--kmer_size", str(KMER), "--genotype_dir", os.path.join(d, tag), "--max_threads",
                                str(n_threads), "--seed", "42"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
            t_all = time.time() - t0
            if g.returncode:
                return {"error": g.stdout[-400:]}
            t_map = t_load = None
            feed = None
            for line in g.stdout.splitlines():
                if "Quasimap (parse + map" in line:
                    t_map = float(line.rsplit(":", 1)[1])
                if "Load data" in line:
                    t_load = float(line.rsplit(":", 1)[1])
                if line.strip().startswith("feed:"):
                    feed = line.strip()
            return dict(parse_and_map_s=t_map, whole_call_s=t_all, index_load_s=t_load, feed=feed)
        runs = []
        for rep in range(2):  # the same call twice (the FASTQ is in the page cache both times); both are reported
            r_ = one_call(threads, f"run{rep}")
            if "error" in r_:
                return r_
            runs.append(r_)
        # Which reader takes a plain FASTQ depends on the host threads the caller grants (gram_main.cpp: plain_fastq_on_device): below
        # 32 the file's bytes go up as they are and the GPU scans and packs them (round 6), from 32 on the host's parallel parser packs
        # them to 2 bits first. Both routes at the thread counts a user is likely to pass (the reference's default is 1):
        n_reads_cli = sum(r.shape[0] for r in batches)
        by_threads = {}
        for nt, env_, label in ((1, {}, "device text feed"), (16, {}, "device text feed"), (1, {"GMX_HOST_FASTQ": "1"}, "host parser"),
                                (16, {"GMX_HOST_FASTQ": "1"}, "host parser"), (threads, {"GMX_DEVICE_FASTQ": "1"}, "device text feed")):
            best_ = None
            for rep in range(2):
                r_ = one_call(nt, f"t{nt}_{label.split()[0]}", env_)
                if "error" in r_:
                    best_ = r_
                    break
                if best_ is None or (r_["parse_and_map_s"] or 1e30) < (best_["parse_and_map_s"] or 1e30):
                    best_ = r_
            key = f"{label}, --max_threads {nt}"
            by_threads[key] = best_ if "error" in best_ else {"parse_and_map_s": best_["parse_and_map_s"], "value": n_reads_cli / best_["parse_and_map_s"], "unit": "reads/s", "whole_call_s": best_["whole_call_s"]}
        # many samples on one index upload (`--samples_list`, round 5): the same FASTQ as 8 samples of one call
        n_smp = 8
        with open(os.path.join(d, "samples.tsv"), "w") as fh:
            for i in range(n_smp):
                fh.write(f"s{i}\t{os.path.join(d, 'multi', f's{i}')}\t{fq}\n")
        t0 = time.time()
        m = subprocess.run([gram, "genotype", "--gram_dir", d, "--samples_list", os.path.join(d, "samples.tsv"), "--ploidy", "haploid",
                            "--kmer_size", str(KMER), "--max_threads", str(threads), "--seed", "42"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        t_multi = time.time() - t0
        multi = {"samples": n_smp, "whole_call_s": t_multi, "per_sample_s": t_multi / n_smp, "rc": m.returncode,
                 "same_files_as_a_call_of_its_own": None}
        if m.returncode == 0:
            same = True
            # (the genotype/ files carry the sample's name; tests/test_gram_cli.py compares all seven files with equal names)
            for name in ("coverage/allele_sum_coverage", "coverage/allele_base_coverage.json", "coverage/grouped_allele_counts_coverage.json",
                         "read_stats.json"):
                a = open(os.path.join(d, "run1", name), "rb").read()
                same = same and all(open(os.path.join(d, "multi", f"s{i}", name), "rb").read() == a for i in (0, n_smp - 1))
            multi["same_files_as_a_call_of_its_own"] = same
        n = sum(r.shape[0] for r in batches)
        best = min(runs, key=lambda r: r["parse_and_map_s"] or 1e30)
        t_map, t_all, t_load, feed = best["parse_and_map_s"], best["whole_call_s"], best["index_load_s"], best["feed"]
        return {"reads": n, "fastq_bytes": os.path.getsize(fq), "host_threads": threads,
                "parse_and_map_s": t_map, "value": n / t_map if t_map else None, "unit": "reads/s",
                "whole_call_s": t_all, "whole_call_reads_per_s": n / t_all, "index_load_s": t_load, "gram_build_s": t_build,
                "feed": feed, "parse_and_map_s_of_both_calls": [r["parse_and_map_s"] for r in runs],
                "whole_call_s_of_both_calls": [r["whole_call_s"] for r in runs], "samples_in_one_call": multi,
                "reader": "host parser (2-bit planes, 40 B per read over the link): --max_threads >= 32; by_threads holds both readers at 1, 16 and this many threads",
                "by_threads": by_threads,
                "note": "plain four-line FASTQ -> coverage files, the call made twice and the faster one quoted; parse_and_map = parser threads (2-bit planes) + H2D + kernels"}


def _container_memory_gib():
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            return int(v) / 2 ** 30
    except OSError:
        pass
    return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2 ** 30


def config_leg(which, n_reads, steps, device, stream):
    """BASELINE.json configs[which] at full size (SURVEY §8(d)'s recipe): index built, `n_reads` reads mapped through the
    kernel-pipeline loop (bytes resident in HBM) and the packed host feed (2-bit stream from page-locked memory), with the
    leading kernels' own durations. The reference's loop being replaced: quasimap.cpp:82-141."""
    import torch
    from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads_2bit, PinnedArray
    from gramtools_amd.synth import chr20_recipe, flat_offsets, genome_recipe_file, pf3d7_recipe
    t0 = time.time()
    if which == 2:
        k, what = 10, "configs[2]: 23.3 Mb random ref + 2000 nested MSA regions (depth <= 3) + 100 k SNPs, k = 10"
        prg, reads_all = pf3d7_recipe(23_300_000, 2000, 100_000, 4 * n_reads, 22)  # (the last leg below maps all of them as ONE batch)
        reads = reads_all[:n_reads]
        ix = Index(prg, k)
    elif which == 3:
        k, what = 14, "configs[3]: 64 444 167 bp random ref + 1.8 M sites (90 % SNP / 10 % indel, 5 % multi-allelic), k = 14"
        prg, reads_all = chr20_recipe(64_444_167, 1_800_000, 4 * n_reads, 32)
        reads = reads_all[:n_reads]
        ix = Index(prg, k)
    else:
        k, what = 14, "configs[4]: 3.1 G bases + 85 M sites (the configs[3] mix), k = 14; index replicated per GPU"
        if os.environ.get("GMX_BENCH_SKIP_C4"):
            return {"skipped": "GMX_BENCH_SKIP_C4 is set"}
        if _container_memory_gib() < 280:
            return {"skipped": f"needs >= 280 GiB of host memory for the index build; this box has {_container_memory_gib():.0f} GiB"}
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gmx_bench_{os.getpid()}.prg")
        _, reads = genome_recipe_file(path, 3_100_000_000, 85_000_000, n_reads, 61)
        reads_all = None
        ix = Index(path, k)
        os.remove(path)
        prg = None
    del prg
    info = ix.info
    t_build = time.time() - t0
    n = reads.shape[0]
    seeds = master_seeds(42, [n])
    offs = flat_offsets(n, reads.shape[1])
    flat = np.ascontiguousarray(reads).reshape(-1)
    qm = Quasimapper(ix, device=device)
    d_r, d_o = torch.from_numpy(flat).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda()
    d_s = torch.from_numpy(np.ascontiguousarray(seeds).view(np.int32).copy()).cuda()

    def loop_kernel(count):
        qm.reset(stream=stream)
        for _ in range(count):
            qm.map_reads_device(d_r, d_o, d_s, n, stream=stream)
        qm.sync()

    loop_kernel(2)                      # (sizes the workspace, warms the clocks)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        loop_kernel(steps)
        rates.append(n * steps / (time.perf_counter() - t0))
    qm.enable_timing(True)
    loop_kernel(3)
    tm = qm.timing()
    qm.enable_timing(False)
    queues = qm.queue_counts()
    big_batch = None
    if reads_all is not None:
        # Every batch has a tail of few-lane kernels; a nested PRG's batch ends with ~2 ms of a few straggler tasks (reads inside MSA regions) whatever its size: the engine
        # takes a whole call of up to max_batch_reads (4 M) as ONE launch there (gmx_feed_chunk), `gram` hands over blocks that size.
        n4 = reads_all.shape[0]
        flat4 = np.ascontiguousarray(reads_all).reshape(-1)
        d_r4, d_o4 = torch.from_numpy(flat4).cuda(), torch.from_numpy(flat_offsets(n4, reads_all.shape[1]).astype(np.int64)).cuda()
        d_s4 = torch.from_numpy(np.ascontiguousarray(master_seeds(42, [n4])).view(np.int32).copy()).cuda()
        r4 = []
        for rep_ in range(4):
            qm.reset(stream=stream)
            qm.sync()
            t0 = time.perf_counter()
            for _ in range(max(2, steps // 2)):
                qm.map_reads_device(d_r4, d_o4, d_s4, n4, stream=stream)
            qm.sync()
            r4.append(n4 * max(2, steps // 2) / (time.perf_counter() - t0))
        st4 = qm.coverage().stats.as_dict()
        big_batch = {"reads_per_batch": n4, "value": float(np.median(r4[1:])), "unit": "reads/s", "runs": [float(r) for r in r4[1:]],
                     "all_reads_mapped": st4["exact_mapped"] >= n4 * max(2, steps // 2),
                     "note": "the same kernel pipeline, 4 x the reads per launch: a batch's tail (a nested PRG's stragglers, the few-lane kernels) is paid once per launch"}
        del d_r4, d_o4, d_s4, flat4
    pk = pack_reads_2bit(flat, offs, uniform_len=reads.shape[1], pinned=True)
    sd = PinnedArray(n, np.uint32)          # page-locked like the stream, read in place by the few reads that draw — as the headline's
    sd.array[:] = seeds                      # (pageable seeds are registered and unregistered by every call, and the call then waits for
    qm.seeds_in_place(True)                  #  its uploads: the copy of batch i + 1 no longer runs beside the kernels of batch i)
    feed = []
    for rep_ in range(4):
        qm.reset()
        qm.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            qm.map_reads_packed(pk, sd.array, use_skip=False)
        qm.sync()
        feed.append(n * steps / (time.perf_counter() - t0))
    st = qm.coverage().stats.as_dict()
    pk.close()
    sd.close()
    L = max(tm["search_launches"], 1)
    kernels = {"gmx_extend_kernel (first pass)": tm["search_ms"] / L}
    for nm, label in (("extend2", "gmx_extend_kernel (first pass over the stragglers)"), ("single", "single-instance coverage kernel"),
                      ("seed", "gmx_seed_kernel / gmx_probe_kernel"), ("filter0", "k-mer filter, pass 0"), ("filter1", "k-mer filter, pass 1")):
        kk = tm["kernels"][nm]
        if kk["launches"]:
            kernels[label] = kk["ms"] / kk["launches"]
    dominant = max(kernels, key=kernels.get)
    # roofline objects (round 6): algorithmic bytes and counter traffic from the configuration's own profile
    # (profiles/round*/config<N>_roofline.json: rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes and the stats build's step
    # counts of the same workload; tools/config_roofline.py), the kernels' durations LIVE from this run's dispatch events
    rj, rj_src = profile_json(f"config{which}_roofline.json")
    roof = []
    for obj in (rj or {}).get("roofline", []):
        kn = obj.get("kernel", "")
        if kn.startswith("gmx_extend_kernel"):
            ms = tm["search_ms"] / L
        elif kn.startswith("gmx_probe_kernel") or kn.startswith("gmx_seed_kernel"):
            ms = tm["kernels"]["seed"]["ms"] / max(tm["kernels"]["seed"]["launches"], 1)
        elif kn.startswith("gmx_cover"):
            ms = tm["kernels"]["single"]["ms"] / max(tm["kernels"]["single"]["launches"], 1)
        else:
            continue
        if not ms:
            continue
        per_launch = obj["alg_bytes_per_read"] * n if "alg_bytes_per_read" in obj else obj.get("alg_bytes_per_task", 0) * 2 * n
        a = per_launch / (ms / 1e3) / 1e9
        roof.append({"kernel": kn, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                     "traffic": obj.get("traffic"), "traffic_over_algorithmic": (obj["traffic"] / per_launch) if obj.get("traffic") and per_launch else None,
                     "avg_launch_ms": ms, "alg_bytes_per_launch": int(per_launch), "alg_bytes_model": obj.get("alg_bytes_model"),
                     "measured": "duration: HIP events attached to this run's dispatches; bytes model and counter traffic: " + str(rj_src)})
    out = {"workload": what, "roofline": roof, "reads_per_step": n, "steps": steps, "symbols": int(info.n_text - 1), "sites": int(info.n_sites),
           "index_bytes": int(info.index_bytes), "is_nested": bool(info.is_nested), "build_and_generate_s": round(t_build, 1),
           "kernel_pipeline": {"value": float(np.median(rates)), "unit": "reads/s", "runs": [float(r) for r in rates]},
           "packed_host_feed": {"value": float(np.median(feed[1:])), "unit": "reads/s", "runs": [float(r) for r in feed[1:]]},
           "kernel_ms": {k_: round(v, 4) for k_, v in kernels.items()}, "dominant_kernel": dominant,
           "batch_ms_kernel_pipeline": n / float(np.median(rates)) * 1e3, "kernel_pipeline_large_batch": big_batch,
           "stats": st, "all_reads_mapped": st["exact_mapped"] >= n * steps and st["all"] == 2 * n * steps,
           "routes": {k_: int(queues[k_]) for k_ in ("mapped", "cover_general", "overflow_probe", "overflow_extend", "inst_mapped", "seed_cursor")}}
    del qm, ix, d_r, d_o, d_s
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU per step (weak scaling)")
    ap.add_argument("--total-reads", type=int, default=0,
                    help="strong scaling: this many reads per step for the WHOLE job, split over the GPUs")
    ap.add_argument("--batches", type=int, default=N_BATCHES, help="distinct batches of reads cycled by the timed loop")
    ap.add_argument("--jobs", type=int, default=0, help="the timed job (exactly --steps steps) is run this many times; value = the median job. "
                                                        "0 (default): as many as it takes for the jobs to add up to 1 s of timed work (5 <= jobs <= 150)")
    ap.add_argument("--configs", default="2,3,4", help="N = 1 side legs: BASELINE configs built at full size (configs[4] runs when the box has >= 280 GiB of host memory: ~6 minutes; "
                                                       "GMX_BENCH_SKIP_C4=1 leaves it out)")
    ap.add_argument("--config-reads", type=int, default=1_000_000, help="reads per step of the --configs legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip kernel_pipeline / sustained / cli_end_to_end / roofline leg")
    ap.add_argument("--cli-reads", type=int, default=4_000_000, help="reads in the FASTQ of the cli_end_to_end leg")
    ap.add_argument("--torch-exchange", action="store_true", help="N > 1: all-reduce through torch.distributed instead of the library")
    ap.add_argument("--planes", action="store_true",
                    help="hand the reads over as bit planes (40 B per 150 bp read) instead of the 2-bit stream (37.5 B)")
    ap.add_argument("--upload-seeds", action="store_true",
                    help="upload the per-read seeds with every batch (default: left in page-locked host memory, read in place)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads, pack_reads_2bit, PinnedArray
    from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads_fast, flat_offsets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the quasimap engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node <gpus>"

    # ---- workload (index replicated; reads sharded by global read index) -----------------------
    t0 = time.time()
    ref = random_ref(GENOME, 1)
    prg, pos, alts, n_alts = snp_prg(ref, N_SITES, 2)
    ix = Index(prg, KMER)
    t_index = time.time() - t0
    strong = args.total_reads > 0
    n = args.total_reads // world if strong else args.reads
    NB = max(1, args.batches)
    t0 = time.time()
    threads = min(os.cpu_count() or 8, 64)
    with ThreadPoolExecutor(max_workers=min(NB, max(1, threads // max(world, 1)))) as ex:
        # batch j of this rank: its own random stream (distinct reads in every batch and on every rank)
        raw = list(ex.map(lambda j: simulate_snp_reads_fast(ref, pos, alts, n_alts, n, READ_LEN, 1000 + 97 * j + rank), range(NB)))
    offs = flat_offsets(n, READ_LEN)
    all_seeds = master_seeds(42, [n * world * NB])      # one master stream for the whole job ...
    batches = []
    for j in range(NB):                                 # ... read i of batch j of rank r is global read (j * world + r) * n + i
        pk = (pack_reads if args.planes else pack_reads_2bit)(raw[j].reshape(-1), offs, uniform_len=READ_LEN,
                                                              threads=max(1, threads // max(world, 1)), pinned=True)
        sd = PinnedArray(n, np.uint32)
        sd.array[:] = all_seeds[(j * world + rank) * n:(j * world + rank + 1) * n]
        batches.append((pk, sd))
    t_reads = time.time() - t0
    reads = raw[0]
    seeds = np.array(batches[0][1].array, copy=True)  # (a copy: the page-locked arrays are freed before the config legs, cpu_baseline runs after them)
    qm = Quasimapper(ix, device=local_rank)
    if not args.upload_seeds:
        # the per-read seeds stay in page-locked host memory: only a read with several equally good mapping classes draws, and
        # the kernels read those few seeds in place (gmx_engine_seeds_in_place) — 40 instead of 44 bytes per read over PCIe
        qm.seeds_in_place(True)
    stream = torch.cuda.current_stream().cuda_stream   # the default stream: the one the host feed launches on
    from gramtools_amd.distributed import allreduce_device_coverage, fused_coverage_tensor, CoverageComm
    exchange = "none"
    comm = cov_t = None
    if world > 1:
        ok = 0
        if not args.torch_exchange:
            try:
                comm = CoverageComm(qm, dist)
                ok = 1
            except Exception as exc:  # RCCL could not be driven from the library on this rank: every rank falls back
                print(f"[rank {rank}] library exchange unavailable ({exc}); using torch.distributed", file=sys.stderr)
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            exchange = "library (gmx_comm_allreduce_coverage: RCCL all-reduce from C++)"
        else:
            if comm is not None:
                comm.close()
                comm = None
            cov_t = fused_coverage_tensor(qm)
            exchange = "torch.distributed all_reduce on the aliased block"

    def exchange_coverage():
        if comm is not None:
            comm.allreduce(stream)
        elif cov_t is not None:
            allreduce_device_coverage(qm, dist, cov_t, stream)

    def fence():
        qm.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def job(steps, first=0):
        """THE job: zeroed accumulators -> `steps` batches from host memory -> one exchange -> coverage arrays on the host."""
        qm.reset()
        for s in range(first, first + steps):
            pk, sd = batches[s % NB]
            qm.map_reads_packed(pk, sd.array, use_skip=False)   # asynchronous: H2D on the copy stream beside the kernels
        exchange_coverage()
        return qm.coverage()                                      # waits, D2H of the accumulator block, gather

    def timed(fn, *a):
        fence()
        t0 = time.perf_counter()
        out = fn(*a)
        fence()
        own = time.perf_counter() - t0
        dt = own
        if world > 1:
            tt = torch.tensor([own], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, own, out

    # ---- side legs first (the device has idled through the index build: they also bring its clocks up) ----
    extras = world == 1 and not args.no_extras
    side = {}
    tm = None
    d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_seeds = torch.from_numpy(np.asarray(seeds).astype(np.int64)).to(torch.int32).cuda()

    def kernel_job(steps):  # rounds 1-2's region: reads resident in HBM, the same batch every step, coverage left in HBM
        qm.reset(stream=stream)
        for _ in range(steps):
            qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
        exchange_coverage()

    # roofline leg: the dominant kernel bracketed by HIP events attached to its dispatch inside the library
    kernel_job(10)
    qm.enable_timing(True)
    fence()
    kernel_job(5)
    fence()
    tm = qm.timing()
    qm.enable_timing(False)
    q1 = qm.queue_counts() if rank == 0 else None
    if extras:
        dtk, _, _ = timed(kernel_job, args.steps)
        side["kernel_pipeline"] = {"value": n * args.steps / dtk, "unit": "reads/s", "ms_per_step": dtk / args.steps * 1e3,
                                   "note": "rounds 1-2's `value`: reads resident in HBM (1 byte per base, gmx_pack_kernel in the "
                                           "pipeline), ONE batch replayed, coverage left in HBM"}

        if len(raw) >= 4:  # the same loop with four batches' reads per launch (a batch's tail of few-lane kernels is paid once per launch)
            n4 = 4 * n
            d_r4 = torch.from_numpy(np.concatenate([r_.reshape(-1) for r_ in raw[:4]])).cuda()
            d_o4 = torch.from_numpy(flat_offsets(n4, READ_LEN).astype(np.int64)).cuda()
            d_s4 = torch.from_numpy(np.asarray(master_seeds(42, [n4])).astype(np.int64)).to(torch.int32).cuda()

            def big_job(steps_):
                qm.reset(stream=stream)
                for _ in range(steps_):
                    qm.map_reads_device(d_r4, d_o4, d_s4, n4, stream=stream)
                exchange_coverage()
            big_job(2)
            dt4, _, _ = timed(big_job, max(2, args.steps // 4))
            side["kernel_pipeline_large_batch"] = {"value": n4 * max(2, args.steps // 4) / dt4, "unit": "reads/s", "reads_per_launch": n4}
            del d_r4, d_o4, d_s4

    # ---- W warm-up steps, then THE timed region: exactly `steps` steps, max over ranks ----
    if args.warmup:
        job(args.warmup)
    # `--jobs` jobs of EXACTLY `steps` steps each, every one between fences (barrier + device synchronisation on both
    # sides, max over ranks); the MEDIAN job is the one quoted (a 15 ms job moves by several per cent on one hiccup)
    # (round 6: `value` rests on >= 1 s of timed jobs — a K = 20 job takes 15 ms and five of them were all the headline stood on;
    #  every job is still EXACTLY K steps between its own fences, the count of jobs is what grows)
    runs = [timed(job, args.steps, args.warmup)]
    n_jobs = args.jobs if args.jobs > 0 else int(min(150, max(5, -(-1.0 // max(runs[0][0], 1e-4)))))
    if world > 1:  # (every rank must run the same number of jobs: rank 0's count)
        nj = torch.tensor([n_jobs], device="cuda")
        dist.broadcast(nj, 0)
        n_jobs = int(nj.item())
    for j in range(1, n_jobs):
        runs.append(timed(job, args.steps, args.warmup + j * args.steps))
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    dt, own, cov = runs[order[len(order) // 2]]
    job_seconds = [r[0] for r in runs]
    st = cov.stats.as_dict() if rank == 0 else None
    per_rank = None
    exchange_ms = None
    if world > 1:
        gathered = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([own], device="cuda", dtype=torch.float64))
        per_rank = [float(g.item()) for g in gathered]
        dte, _, _ = timed(exchange_coverage)   # the exchange alone (the block holds the job's totals: sums again, unused)
        exchange_ms = dte * 1e3

    total_reads = n * world * args.steps
    value = total_reads / dt
    out = None
    if rank == 0:
        search_s = tm["search_ms"] / 1e3 / max(tm["search_launches"], 1)
        reads_per_launch = tm["reads"] / max(tm["search_launches"], 1)
        achieved = B_DESIGN_PER_READ * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0
        k_seed = max(KMER, int(ix.info.kmer_size2))
        b_nominal_kernel = 128 * (READ_LEN - k_seed) + READ_LEN  # the part of the nominal figure this kernel is credited with
        sq, sq_src = profile_json("sq_extend.json")
        sq = sq or {}
        traffic, traffic_src = measured_traffic("gmx_extend_kernel")
        h2d_per_read = (8 * ((READ_LEN + 31) // 32) if args.planes else READ_LEN / 4) + (4 if args.upload_seeds else 0)
        h2d_gbs = h2d_per_read * value / world / 1e9
        # ---- one roofline object per leading kernel; durations measured live (events attached to each dispatch) ----
        def k_ms(name):
            kk = tm["kernels"][name]
            return kk["ms"] / kk["launches"] if kk["launches"] else None
        n_tasks_l = 2 * reads_per_launch
        n_dead0 = q1["dead"] if q1 else None
        n_mapped = q1["mapped"] if q1 else reads_per_launch
        ext = {"kernel": "gmx_extend_kernel<false,1>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "frac_by_counter_traffic": (traffic / search_s / 1e9 / HBM_PEAK_GBS) if traffic and search_s > 0 else None,
               "traffic_source": f"{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 1 M reads per launch)",
               "alg_bytes_per_read": B_DESIGN_PER_READ,
               "alg_bytes_model": "text-form states: 32 B of PRG per 64 symbols, SNP sites resolved inside the record; "
                                  "a 16 B sub-record only for sites that straddle a record end (DESIGN.md §8)",
               "reads_per_launch": reads_per_launch, "avg_launch_ms": search_s * 1e3,
               "measured": "HIP events attached to the dispatch (hipExtLaunchKernelGGL), reads resident in HBM leg",
               "what_bounds_it": "instruction issue of the wave loop and the slowest lane of each wave, not HBM bandwidth",
               "issue": {"valu_busy": sq.get("valu_busy"), "active_lane_share": sq.get("active_lane_share"),
                         "frac": (sq.get("valu_busy") or 0) * (sq.get("active_lane_share") or 0) or None,
                         "iterations_per_wave": sq.get("iterations_per_wave"),
                         "heavy_steps_per_lane": sq.get("heavy_steps_per_lane"),
                         "source": f"{sq_src} (rocprofv3 --pmc SQ counters + GMX_LOOP_STATS build)"},
               "nominal": {"bytes_per_read": b_nominal_kernel, "bytes_per_read_whole_path": B_NOMINAL_PER_READ,
                           "achieved": b_nominal_kernel * reads_per_launch / search_s / 1e9 if search_s > 0 else 0.0,
                           "note": "SURVEY §8(d) prices a 128 B rank block per base (the reference's algorithm); "
                                   "exceeds the HBM peak because the kernel does not move those bytes"}}
        kernel_rooflines = [ext]

        def add_kernel(name, ms, bytes_per_launch, model, bounded_by, traffic_name=None):
            if not ms:
                return
            tr, tr_src = measured_traffic(traffic_name) if traffic_name else (None, None)
            a = bytes_per_launch / (ms / 1e3) / 1e9
            kernel_rooflines.append({"kernel": name, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": a / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src, "avg_launch_ms": ms,
                                     "alg_bytes_per_launch": int(bytes_per_launch), "alg_bytes_model": model,
                                     "what_bounds_it": bounded_by})
        add_kernel("gmx_cover_jump_kernel", k_ms("single"), n_mapped * (32 + 4 + 2 * 32 + 2 * 8),
                   "per mapped read: compact record 32 + task id 4 + a 32 B geometry record and one 8 B atomic per site crossed (2 at configs[1])",
                   "the memory system's rate of scattered accesses (a sector + an atomic per site crossed), not bandwidth",
                   "gmx_cover_jump_kernel")
        if n_dead0:
            add_kernel("k-mer filter, pass 0 (gmx_filter_lds_kernel)", k_ms("filter0"), n_dead0 * (48 + 4) + 256 * 131072,
                       "per dead task: read planes 48 + queue entry 4; + the 128 KB presence bitmap staged into LDS by each of 256 workgroups (from L2)",
                       "LDS bitmap probes and VALU (141 k-mers per dead task), one 1024-thread workgroup per CU; decides only which COUNTER "
                       "(missing_kmer / no_extension, quasimap.cpp:212-225) a task without states lands in; runs on a side stream",
                       "gmx_filter_lds_kernel")
        add_kernel("gmx_seed_kernel", k_ms("seed"), n_tasks_l * (16 + 8) + reads_per_launch * 12 + (n_dead0 or 0) * 4,
                   "per task: the read's last plane pair 16 + seed directory entry 8; per alive task 12 (queue entry + entry copy), per dead task 4",
                   "2 M scattered 8-byte entries of a 537 MB table per launch", "gmx_seed_kernel")
        # which kernel leads: the arg-max of the LIVE timers of this run (round 5's line named gmx_extend_kernel by a literal while its own
        # timers showed the filter's first pass longer). Two answers, both stated: the longest kernel on the batch's main chain
        # (unpack -> seed -> extend -> extend<2> -> single-instance coverage: what a batch cannot be shorter than) and the longest
        # dispatch on any stream (the k-mer filter's passes run on side streams beside that chain: tools/kab.py, round 6 — without
        # the filter a step of the kernel pipeline is 4-6 % shorter, so its duration is mostly waiting for a CU's LDS, not work).
        live = {"gmx_extend_kernel<false,1>": (search_s * 1e3, "main"), "gmx_seed_kernel": (k_ms("seed"), "main"),
                "gmx_cover_jump_kernel": (k_ms("single"), "main"), "gmx_extend_kernel<false,2> (stragglers)": (k_ms("extend2"), "main"),
                "gmx_unpack2_kernel": (k_ms("unpack"), "main"), "gmx_filter_lds_kernel pass 0": (k_ms("filter0"), "side stream 1"),
                "gmx_filter_lds_kernel pass 1": (k_ms("filter1"), "side stream 2")}
        live = {k_: v_ for k_, v_ in live.items() if v_[0]}
        dominant = {"longest_on_the_main_chain": max((k_ for k_ in live if live[k_][1] == "main"), key=lambda k_: live[k_][0]),
                    "longest_dispatch_on_any_stream": max(live, key=lambda k_: live[k_][0]),
                    "live_ms": {k_: {"ms": round(v_[0], 4), "stream": v_[1]} for k_, v_ in live.items()},
                    "measured": "HIP events attached to each dispatch in this run (reads-resident leg, 5 timed batches)"}
        out = {
            "metric": "150bp reads quasimapped/sec (whole node); bit-exact coverage",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "value_is": "SURVEY §8(d) region: host buffers in (2-bit planes, page-locked) -> H2D -> kernels -> (N > 1: one exchange) "
                        "-> D2H: coverage arrays final on the host; distinct reads every step",
            "timed_region": "changed in round 3: rounds 1-2 timed the kernel pipeline only (reads resident in HBM, one batch replayed, "
                            "coverage left in HBM; that figure is the key `kernel_pipeline`), and N > 1 jobs do ONE exchange at the "
                            "end of the job, not one per step: values of different rounds are not comparable",
            "config": {"workload": "configs[1]: M. tuberculosis scale, 4411532 bp random ref + 60000 SNP PRG, k=10, "
                                   f"{n} x 150 bp reads per GPU per step, fwd+rc, {NB} distinct batches cycled, reads handed over as "
                                   + ("2-bit planes" if args.planes else "a 2-bit stream (37.5 B per read)") + " in page-locked host memory"
                                   + ("" if args.upload_seeds else ", per-read seeds read in place from page-locked host memory")
                                   + "; no skip plane (the synthetic reads hold no N: gmx.h takes a null skip pointer as 'none skipped')",
                       "reads_per_gpu": n, "read_len": READ_LEN, "kmer_size": KMER, "distinct_batches": NB,
                       "h2d_bytes_per_read": h2d_per_read,
                       "parallelism": f"reads sharded x{world} by global read index, index replicated, one RCCL all-reduce of the "
                                      "coverage block per JOB (after the last step), then D2H",
                       "exchange": exchange, "index_build_s": round(t_index, 2), "reads_generation_s": round(t_reads, 2),
                       "index_bytes": int(ix.info.index_bytes)},
            "h2d_rate_GBps": h2d_gbs,
            "jobs": {"n": len(job_seconds), "total_seconds": float(sum(job_seconds)), "value_is": "the median job (each job: exactly `steps` steps between fences; enough jobs for >= 1 s of timed work)",
                     "min_value": total_reads / max(job_seconds), "max_value": total_reads / min(job_seconds),
                     "p10_p90_values": [total_reads / float(np.percentile(job_seconds, 90)), total_reads / float(np.percentile(job_seconds, 10))],
                     "first_values": [total_reads / x for x in job_seconds[:8]]},
            "roofline": {
                # what bounds the STEP: the host link. 37.5 B per read cross PCIe once; the kernels of a batch take less than
                # half of the step (kernel_pipeline), the rest of the time the GPU waits for the next batch's bytes.
                "bound": "pcie", "achieved": h2d_gbs, "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": h2d_gbs / PCIE_PEAK_GBS,
                "traffic": int(h2d_per_read * n),
                "traffic_is": "bytes uploaded per step and GPU (exact: reads x h2d_bytes_per_read; no PMC counter sees the host link)",
                "peak_source": "MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec), per direction",
                "step_ms": dt / args.steps * 1e3,
                "kernels_ms_per_step": side.get("kernel_pipeline", {}).get("ms_per_step"),
                "dominant_kernel": dominant["longest_on_the_main_chain"],
                "dominant_kernel_is": dominant,
                "kernels": kernel_rooflines,
            },
            "stats_job": st,
        }
        out.update(side)
        if per_rank is not None:
            out["per_rank_s"] = per_rank
            out["exchange_ms"] = exchange_ms
    if extras:
        # ---- sustained: the same loop for >= 1 s --------------------------------------------------------------
        per_step = dt / args.steps
        k = max(int(1.25 / per_step), args.steps)
        dts, _, _ = timed(job, k)
        out["sustained"] = {"seconds": dts, "steps": k, "reads": k * n, "value": k * n / dts, "unit": "reads/s"}
        # ---- the executable on a FASTQ file -----------------------------------------------------------------------
        n_cli = max(1, min(NB, -(-args.cli_reads // n)))
        out["cli_end_to_end"] = cli_end_to_end(prg, raw[:n_cli], min(os.cpu_count() or 8, 64))
        # ---- the same reads as a BGZF FASTQ, decoded on the device ---------------------------------------------------
        try:
            out["bgzf_device_feed"] = bgzf_device_feed(ix, np.concatenate(raw[:3]) if len(raw) >= 3 else raw[0], master_seeds(42, [3 * n if len(raw) >= 3 else n]))  # (3 M reads: two full chunks and a part)
        except Exception as exc:  # a leg must not cost the headline
            out["bgzf_device_feed"] = {"error": repr(exc)[:300]}
        # ---- the other BASELINE configurations at full size (their own index, 1 M reads per step) ---------------------
        del d_reads, d_offs, d_seeds
        for pk_, sd_ in batches:
            pk_.close()
        batches.clear()
        del qm
        torch.cuda.empty_cache()
        out["configs"] = {}
        for which in [int(x) for x in args.configs.split(",") if x.strip()]:
            try:
                out["configs"][str(which)] = config_leg(which, args.config_reads, 6, local_rank, stream)
            except Exception as exc:  # a leg must not cost the headline
                out["configs"][str(which)] = {"error": repr(exc)[:300]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prg, reads, np.asarray(seeds))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
