"""Parity fuzz on flat PRGs: random site densities, allele lengths (0..max_len bases), allele counts (2..9: dense counters and the
append log), adjacent sites, ragged read lengths, both strands — the HIP path against the oracle, bit-exact, case after case
until the time is up. Usage: python tools/fuzz_parity.py [SECONDS=120] [FIRST_SEED=0] [new]
`new` (round 6): the reads go through the packed feed in launches of 600 reads with two launches in flight (GMX_TWIN=1), every
second case with the seed cursor and its screening side table forced on (GMX_SEED_CURSOR=1), every fourth without the side table."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from common import canonical_cov, flatten_reads, oracle_map  # noqa: E402
from gramtools_amd import Index, Quasimapper  # noqa: E402
from gramtools_amd.synth import mixed_variant_prg, random_ref, simulate_haplotype_reads  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
new_paths = len(sys.argv) > 3 and sys.argv[3] == "new"
if new_paths:
    from gramtools_amd import pack_reads  # noqa: E402
t0, cases, jump_sites, sites = time.time(), 0, 0, 0
while time.time() - t0 < secs:
    rng = np.random.default_rng(1000 + seed)
    G = int(rng.integers(5_000, 40_000))
    density = [4, 6, 10, 25, 60, 150][int(rng.integers(0, 6))]  # bases per site
    max_len = [1, 3, 6, 12, 30][int(rng.integers(0, 5))]
    ref = random_ref(G, 7 * seed + 1)
    prg, st = mixed_variant_prg(ref, max(G // density, 3), 7 * seed + 2, max_alleles=int(rng.integers(3, 10)), max_len=max_len,
                                adjacent_prob=float(rng.choice([0.0, 0.05, 0.3])))
    k = int(rng.choice([5, 7, 9, 11]))
    lo = int(rng.integers(k, 120))
    reads = [r for r in simulate_haplotype_reads(ref, st, 2500, lo, lo + int(rng.integers(1, 250)), 7 * seed + 3) if len(r) >= k]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 2654435761 + seed).astype(np.uint32)
    rng_mode = seed % 2
    want = oracle_map(prg, k, reads, seeds, rng_mode=rng_mode, threads=8)
    ix = Index(prg, k)
    flat, offs = flatten_reads(reads)
    if new_paths:
        os.environ["GMX_TWIN"] = "1"
        os.environ.pop("GMX_SEED_CURSOR", None)
        os.environ.pop("GMX_NO_SEED_SIDE", None)
        if seed % 2:
            os.environ["GMX_SEED_CURSOR"] = "1"
        if seed % 4 == 3:
            os.environ["GMX_NO_SEED_SIDE"] = "1"
        qm = Quasimapper(ix, rng_mode=rng_mode, max_batch_reads=600)
        pk = pack_reads(flat, offs, pinned=True)
        qm.map_reads_packed(pk, seeds)
    else:
        qm = Quasimapper(ix, rng_mode=rng_mode)
        qm.map_reads(flat, offs, seeds)
    got = canonical_cov(qm.coverage())
    if got != want:
        print(f"MISMATCH at seed {seed}: G {G}, a site per {density} bases, max_len {max_len}, k {k}, reads from {lo} bases, stats {got['stats']} / {want['stats']}", flush=True)
        sys.exit(1)
    if new_paths:
        pk.close()
    jump_sites += ix.info.n_jump_sites
    sites += ix.info.n_sites
    cases += 1
    seed += 1
print(f"{cases} cases bit-exact in {time.time() - t0:.0f} s (next seed {seed}); {jump_sites} of {sites} sites with geometry records")
