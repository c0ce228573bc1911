"""BASELINE.json configs[4] as a builder-and-mapping run on one MI355X: a whole-genome-sized PRG (3.1 G random bases,
85 M SNP sites -> 3.46 G symbols; the recipe of SURVEY §8d at genome scale), index built (GMX_BUILD_TRACE phases),
cached, reloaded, uploaded, and error-free reads mapped with the size-independent properties of tools/scale_check.py.
Usage: python tools/scale_check_config4.py [GENOME=3100000000] [N_SITES=85000000] [K=14] [N_READS=1000000]"""
import os
import resource
import shutil
import sys
import time

import numpy as np

os.environ.setdefault("GMX_BUILD_TRACE", "1")
sys.path.insert(0, ".")
import torch  # noqa: E402

torch.cuda.init()
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads_fast, snp_prg  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 85_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 14
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000


def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def say(*a):
    print(f"[{time.time() - T0:8.1f} s, peak RSS {rss_gb():6.1f} GB]", *a, flush=True)


T0 = time.time()
say(f"host: {os.cpu_count()} hardware threads, {os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES') / 1e9:.0f} GB RAM; "
    f"device: {torch.cuda.get_device_name(0)}, {torch.cuda.get_device_properties(0).total_memory / 1e9:.0f} GB")
ref = random_ref(G, 1)
prg, pos, alts, n_alts = snp_prg(ref, n_sites, 2)
say(f"PRG: {prg.size} symbols, {n_sites} sites over {G} bases")
reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, n_reads, 150, 1000)
say(f"{n_reads} error-free 150 bp reads simulated")
del ref
t0 = time.time()
ix = Index(prg, k)
info = ix.info
say(f"index built in {time.time() - t0:.1f} s: {info.index_bytes / 1e9:.1f} GB, k = {k}, k2 = {info.kmer_size2}, "
    f"{info.n_sites} sites, {info.n_inline_sites} inline")
tmp = os.environ.get("TMPDIR", "/tmp")
free = shutil.disk_usage(tmp).free
cache = os.path.join(tmp, "gmx_config4.idx")
prg_path = os.path.join(tmp, "gmx_config4.prg")
if free > 3 * info.index_bytes:
    t0 = time.time()
    ix.save(cache)
    np.asarray(prg, dtype="<u4").tofile(prg_path)
    say(f"cache written in {time.time() - t0:.1f} s: {os.path.getsize(cache) / 1e9:.1f} GB")
    ix.close()
    del ix
    t0 = time.time()
    ix = Index(prg_path, k, cache=cache)
    assert ix.from_cache
    say(f"cache loaded (checksummed) in {time.time() - t0:.1f} s")
    os.remove(cache)
    os.remove(prg_path)
else:
    say(f"cache round trip skipped: {free / 1e9:.0f} GB free under {tmp}")
del prg
t0 = time.time()
qm = Quasimapper(ix)
say(f"engine created (index in HBM: {torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9:.1f} GB of the device in use) in {time.time() - t0:.1f} s")
seeds = master_seeds(42, [n_reads])
offs = flat_offsets(n_reads, 150)
flat = reads.reshape(-1)
t0 = time.time()
try:
    qm.map_reads(flat, offs, seeds)
    qm.sync()
except Exception as exc:
    print("FAILED:", exc, "\nqueues of the last batch:", qm.queue_counts(), flush=True)
    raise
dt = time.time() - t0
fwd = qm.coverage()
st = fwd.stats.as_dict()
say(f"mapped {n_reads} reads in {dt:.2f} s ({n_reads / dt / 1e6:.2f} M reads/s, host buffers): {st}")
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
a_sum, g_sum, pb_sum = int(fwd.raw_allele_sum.astype(np.int64).sum()), int(fwd.raw_grouped.astype(np.int64).sum()), int(fwd.raw_per_base.astype(np.int64).sum())
assert a_sum >= g_sum > 0
say(f"properties hold: counter identity, every read mapped, strand symmetry, repeatability; allele-sum total {a_sum}, "
    f"grouped total {g_sum}, per-base total {pb_sum}")
