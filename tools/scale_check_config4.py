"""BASELINE.json configs[4] as a builder-and-mapping run on one MI355X: SURVEY §8(d)'s recipe — 3.1 G random bases, 85 M
sites in the configs[3] mix (gramtools_amd.synth.genome_recipe_file) —, index built (GMX_BUILD_TRACE phases with resident
memory), uploaded, and error-free reads mapped with the size-independent properties of tools/scale_check.py, then the
packed host feed and the kernel pipeline timed. The GPU boxes give a container 300 GiB of host
memory and 16 cores of CPU time (cgroup limits), so the number of sites is what the HOST memory of the builder allows,
not what the device could hold; a guard thread ends the process cleanly before the limit (a box that runs out of memory
is lost). A scaled-down run (TRIAL bases) comes first and its peak memory is extrapolated.
Usage: python tools/scale_check_config4.py [GENOME=3100000000] [N_SITES=85000000] [K=14] [N_READS=1000000] [TRIAL=200000000]"""
import gc
import os
import resource
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GMX_BUILD_TRACE", "1")
sys.path.insert(0, ".")
import torch  # noqa: E402

torch.cuda.init()
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, genome_recipe_file  # noqa: E402
from gramtools_amd import pack_reads_2bit  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 85_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 14
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
trial = int(sys.argv[5]) if len(sys.argv) > 5 else 200_000_000
LIMIT_GB = float(os.environ.get("GMX_RSS_LIMIT_GB", "285"))
T0 = time.time()


def rss_now_gb():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e9


def peak_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def say(*a):
    print(f"[{time.time() - T0:8.1f} s, RSS {rss_now_gb():6.1f} GB, peak {peak_gb():6.1f} GB]", *a, flush=True)


def guard():
    while True:
        if rss_now_gb() > LIMIT_GB:
            print(f"ABORT: resident memory above {LIMIT_GB} GB (the container's limit is 300 GiB)", flush=True)
            os._exit(3)
        time.sleep(0.2)


threading.Thread(target=guard, daemon=True).start()


def run(G, n_sites, n_reads, label):
    say(f"== {label}: {G} bases, {n_sites} sites (90 % SNP / 10 % 1-10 bp indels, 5 % with 3-4 alleles), k = {k}")
    tmp = os.environ.get("TMPDIR", "/tmp")
    prg_path = os.path.join(tmp, "gmx_config4.prg")
    n_symbols, reads = genome_recipe_file(prg_path, G, n_sites, n_reads, 61 if G > 10 ** 9 else 51)
    gc.collect()
    say(f"PRG written: {n_symbols} symbols ({os.path.getsize(prg_path) / 1e9:.1f} GB); {n_reads} error-free 150 bp reads simulated")
    t0 = time.time()
    before = peak_gb()
    ix = Index(prg_path, k)
    os.remove(prg_path)
    info = ix.info
    say(f"index built in {time.time() - t0:.1f} s: {info.index_bytes / 1e9:.1f} GB, k2 = {info.kmer_size2}, {info.n_sites} sites, "
        f"{info.n_inline_sites} inline, {info.n_seed_words / 1e9:.2f} G words of multi-state entries on units of 2^{info.seed_shift}; "
        f"peak memory of the process so far {peak_gb():.1f} GB (before the build {before:.1f})")
    t0 = time.time()
    qm = Quasimapper(ix)
    free_b, total_b = torch.cuda.mem_get_info()
    say(f"engine created in {time.time() - t0:.1f} s: {(total_b - free_b) / 1e9:.1f} GB of the device's {total_b / 1e9:.0f} GB in use")
    seeds = master_seeds(42, [n_reads])
    offs = flat_offsets(n_reads, 150)
    flat = reads.reshape(-1)
    t0 = time.time()
    qm.map_reads(flat, offs, seeds)
    qm.sync()
    dt = time.time() - t0
    fwd = qm.coverage()
    st = fwd.stats.as_dict()
    say(f"mapped {n_reads} reads in {dt:.2f} s ({n_reads / dt / 1e6:.2f} M reads/s from host buffers): {st}")
    print("queues of the last batch:", qm.queue_counts(), flush=True)
    assert st["all"] == 2 * n_reads and st["skipped"] == 0
    assert st["exact_mapped"] >= n_reads, "every error-free read maps in at least one orientation"
    assert st["all"] == st["skipped"] + st["missing_kmer"] + st["no_extension"] + st["exact_mapped"]
    t0 = time.time()
    qm.reset()
    qm.map_reads(flat, offs, seeds)
    qm.sync()
    dt = time.time() - t0
    again = qm.coverage()
    assert (fwd.raw_allele_sum == again.raw_allele_sum).all() and (fwd.raw_per_base == again.raw_per_base).all() and (fwd.raw_grouped == again.raw_grouped).all()
    say(f"second pass of the same reads: {dt:.2f} s ({n_reads / dt / 1e6:.2f} M reads/s), identical coverage")
    rc = np.ascontiguousarray((5 - reads[:, ::-1]).astype(np.uint8)).reshape(-1)
    qm.reset()
    qm.map_reads(rc, offs, seeds)
    back = qm.coverage()
    assert (fwd.raw_allele_sum == back.raw_allele_sum).all() and (fwd.raw_per_base == back.raw_per_base).all()
    assert (fwd.raw_grouped == back.raw_grouped).all()
    a_sum, g_sum = int(fwd.raw_allele_sum.astype(np.int64).sum()), int(fwd.raw_grouped.astype(np.int64).sum())
    assert a_sum >= g_sum > 0
    say(f"properties hold: counter identity, every read mapped, strand symmetry, repeatability; allele-sum total {a_sum}, grouped total {g_sum}")
    # the production feed (2-bit stream from page-locked memory) and the kernel pipeline (reads resident in HBM as bytes)
    pk = pack_reads_2bit(flat, offs, uniform_len=150, pinned=True)
    qm.reset()
    qm.map_reads_packed(pk, seeds)
    packed = qm.coverage()
    assert (fwd.raw_allele_sum == packed.raw_allele_sum).all() and (fwd.raw_per_base == packed.raw_per_base).all() and (fwd.raw_grouped == packed.raw_grouped).all()
    for rep in range(2):
        qm.reset()
        qm.sync()
        t0 = time.time()
        for _ in range(5):
            qm.map_reads_packed(pk, seeds)
        qm.sync()
        dt = (time.time() - t0) / 5
    say(f"packed host feed (2-bit stream, page-locked): {dt * 1e3:.2f} ms per {n_reads} reads = {n_reads / dt / 1e6:.1f} M reads/s; identical coverage")
    d_r, d_o, d_s = torch.from_numpy(flat).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(seeds.view(np.int32)).cuda()
    for rep in range(2):
        qm.reset()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            qm.map_reads_device(d_r, d_o, d_s, n_reads)
        qm.sync()
        dt = (time.time() - t0) / 5
    say(f"kernel pipeline (bytes resident in HBM): {dt * 1e3:.2f} ms per {n_reads} reads = {n_reads / dt / 1e6:.1f} M reads/s")
    pk.close()
    qm.close()
    ix.close()
    del qm, ix, fwd, again, back, packed
    gc.collect()
    return n_symbols


say(f"host: {os.cpu_count()} hardware threads; cgroup cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '?'}, "
    f"memory.max {open('/sys/fs/cgroup/memory.max').read().strip() if os.path.exists('/sys/fs/cgroup/memory.max') else '?'}; "
    f"device: {torch.cuda.get_device_name(0)}, {torch.cuda.get_device_properties(0).total_memory / 1e9:.0f} GB")
if trial:
    base = peak_gb()
    run(trial, int(n_sites * trial / G), min(n_reads, 200_000), "trial")
    grow = (peak_gb() - base) * G / trial
    say(f"trial peak {peak_gb():.1f} GB -> extrapolated peak of the full run: {base + grow:.0f} GB (limit {LIMIT_GB:.0f})")
    if base + grow > LIMIT_GB:
        say("the full run would not fit: stopping here")
        sys.exit(4)
run(G, n_sites, n_reads, "configs[4] scale")
