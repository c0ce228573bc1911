"""Larger configurations of BASELINE.json (configs[3] recipe: chr20 scale, k = 14) as a scale check: index build
time and size, mapping rate, and size-independent properties (error-free reads all map in exactly one
orientation class, counter identities, strand symmetry). Usage: python tools/scale_check.py GENOME N_SITES K N_READS [repeats]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()  # before the engine: the process must use torch's HIP runtime instance

sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

G, n_sites, k, n_reads = (int(x) for x in sys.argv[1:5])
t0 = time.time()
ref = random_ref(G, 1)
rep_arg = [a for a in sys.argv[5:] if a.startswith("repeats")]
if rep_arg:  # 5 % (or repeats=FRACTION) of the reference replaced by 10 copies each of 1-5 kb segments
    rng = np.random.default_rng(5)
    frac = float(rep_arg[0].split("=")[1]) if "=" in rep_arg[0] else 0.05
    budget = int(ref.size * frac)
    while budget > 0:
        seg = int(rng.integers(1000, 5001))
        src = int(rng.integers(0, ref.size - seg))
        piece = ref[src:src + seg].copy()
        for _ in range(10):
            dst = int(rng.integers(0, ref.size - seg))
            ref[dst:dst + seg] = piece
        budget -= 10 * seg
prg, pos, alts, n_alts = snp_prg(ref, n_sites, 2, multi_allelic_frac=0.05)
print(f"PRG {prg.size} symbols, {n_sites} sites ({time.time() - t0:.1f} s)", flush=True)
t0 = time.time()
ix = Index(prg, k)
info = ix.info
print(f"index: {time.time() - t0:.1f} s, {info.index_bytes / 1e9:.2f} GB", flush=True)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n_reads, 150, 1000)
seeds = master_seeds(42, [n_reads])
offs = flat_offsets(n_reads, 150)
qm = Quasimapper(ix)
t0 = time.time()
try:
    qm.map_reads(reads.reshape(-1), offs, seeds)
except Exception as exc:  # capacity diagnostics
    print("FAILED:", exc, "\nqueues of the last batch:", qm.queue_counts(), flush=True)
    raise
fwd = qm.coverage()
dt = time.time() - t0
st = fwd.stats.as_dict()
print(f"mapped {n_reads} reads in {dt:.2f} s (host buffers, PCIe-inclusive): {st}", flush=True)
print("queues of the last batch:", qm.queue_counts(), flush=True)
assert st["all"] == 2 * n_reads and st["skipped"] == 0
assert st["exact_mapped"] >= n_reads, "every error-free read maps in at least one orientation"
assert st["all"] == st["skipped"] + st["missing_kmer"] + st["no_extension"] + st["exact_mapped"]
# strand symmetry: the reverse complements give identical coverage
rc = (5 - reads[:, ::-1]).astype(np.uint8)
qm2 = Quasimapper(ix)
qm2.map_reads(np.ascontiguousarray(rc).reshape(-1), offs, seeds)
back = qm2.coverage()
assert (fwd.raw_allele_sum == back.raw_allele_sum).all() and (fwd.raw_per_base == back.raw_per_base).all()
assert (fwd.raw_grouped == back.raw_grouped).all()
assert int(fwd.raw_allele_sum.sum()) >= int(fwd.raw_grouped.sum()) > 0
print("properties hold: counter identity, all reads mapped, strand symmetry, allele-sum >= grouped", flush=True)
# device-resident rate (the bench's timed region, at this configuration)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
qm3 = Quasimapper(ix)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    qm3.reset(stream=stream)  # queued like everything else (bench.py's step)
    qm3.map_reads_device(d_reads, d_offs, d_seeds, n_reads, stream=stream)
qm3.sync()
t0 = time.perf_counter()
steps = 10
for _ in range(steps):
    qm3.reset(stream=stream)
    qm3.map_reads_device(d_reads, d_offs, d_seeds, n_reads, stream=stream)
qm3.sync()
dt = (time.perf_counter() - t0) / steps
print(f"device-resident: {dt * 1e3:.2f} ms per {n_reads} reads = {n_reads / dt / 1e6:.0f} M reads/s", flush=True)
