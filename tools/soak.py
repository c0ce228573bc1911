"""Soak: engines created and destroyed in a loop (device memory must come back), then 20 s of the bench loop with the
error word checked every 2 000 batches. Usage: python tools/soak.py [SECONDS=20]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
import bench  # noqa: E402
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ref = random_ref(bench.GENOME, 1)
prg, pos, alts, n_alts = snp_prg(ref, bench.N_SITES, 2)
ix = Index(prg, bench.KMER)
n = 1 << 20
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
seeds = master_seeds(42, [n])
offs = flat_offsets(n, 150)
frees = []
for i in range(12):
    qm = Quasimapper(ix)
    qm.map_reads(reads[:200000].reshape(-1), offs[:200001], seeds[:200000])
    st = qm.coverage().stats.as_dict()
    assert st["exact_mapped"] == 200000, st
    qm.close()
    torch.cuda.synchronize()
    frees.append(torch.cuda.mem_get_info()[0])
print("device memory free after each of 12 engines (GB):", " ".join(f"{f / 1e9:.3f}" for f in frees))
assert frees[1] - frees[-1] < 64 << 20, "engines leak device memory"  # (the first one pays the runtime's one-time allocations)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
qm = Quasimapper(ix)
stream = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
batches = 0
while time.perf_counter() - t0 < secs:
    qm.reset(stream=stream)
    for _ in range(2000):
        qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
    qm.sync()
    batches += 2000
    st = qm.coverage().stats.as_dict()
    assert st["all"] == 2 * n * 2000 and st["exact_mapped"] == n * 2000, st
dt = time.perf_counter() - t0
print(f"{batches} batches of {n} reads in {dt:.1f} s = {batches * n / dt / 1e9:.2f} G reads/s; counters exact after every 2 000 batches")

# ---- the host feeds of round 3: planes and the 2-bit stream from page-locked memory, seeds uploaded / read in place, engines
# created and destroyed around them (upload slots, copy stream and page-locked blocks must come back too) ----
from gramtools_amd import pack_reads, pack_reads_2bit, PinnedArray  # noqa: E402

want = None
free_after = []
for rnd in range(6):
    for packer in (pack_reads, pack_reads_2bit):
        pk = packer(reads.reshape(-1), offs, uniform_len=150, pinned=True)
        sd = PinnedArray(n, np.uint32)
        sd.array[:] = seeds
        q = Quasimapper(ix)
        q.seeds_in_place(rnd % 2 == 1)
        t1 = time.perf_counter()
        for _ in range(50):
            q.map_reads_packed(pk, sd.array, use_skip=False)
        cov = q.coverage()
        dt1 = time.perf_counter() - t1
        st = cov.stats.as_dict()
        assert st["all"] == 2 * n * 50 and st["exact_mapped"] == n * 50, st
        sums = (int(cov.raw_allele_sum.astype(np.int64).sum()), int(cov.raw_per_base.astype(np.int64).sum()), int(cov.raw_grouped.astype(np.int64).sum()))
        assert want is None or sums == want, (sums, want)  # every feed, every round: the same coverage totals
        want = sums
        q.close()
        pk.close()
        sd.close()
        torch.cuda.synchronize()
        free_after.append(torch.cuda.mem_get_info()[0])
        print(f"round {rnd} {packer.__name__:16s} seeds {'in place' if rnd % 2 else 'uploaded'}: {50 * n / dt1 / 1e9:.2f} G reads/s host-inclusive, "
              f"totals {sums}, device memory free {torch.cuda.mem_get_info()[0] / 1e9:.3f} GB", flush=True)
leak = free_after[0] - free_after[-1]  # (the first engine with a host feed pays the runtime's one-time allocations: ~0.5 GB)
print(f"device memory not returned between the first and the twelfth engine with host feeds: {leak / 1e6:.1f} MB")
assert leak < 64 << 20
