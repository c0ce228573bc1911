"""Bit-exact check at the full size of BASELINE.json configs[1] (4.4 Mb + 60 000 SNP sites, k = 10): the first N reads
of the bench workload through the HIP path and through the oracle, every counter compared.
With --repeats, 5 % of the reference is first replaced by 10 copies each of 1-5 kb segments, so that reads
inside the copies have several mapping instances and the seeded selection decides what is recorded.
Usage: python tools/parity_config1.py [N_READS] [--repeats]   (N = 400000: about 25 s of oracle on 256 host threads)"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402
from common import canonical_cov, oracle_map  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 400_000
ref = random_ref(4_411_532, 1)
if "--repeats" in sys.argv:
    rng = np.random.default_rng(5)
    budget = ref.size // 20
    while budget > 0:
        seg = int(rng.integers(1000, 5001))
        src = int(rng.integers(0, ref.size - seg))
        piece = ref[src:src + seg].copy()
        for _ in range(10):
            dst = int(rng.integers(0, ref.size - seg))
            ref[dst:dst + seg] = piece
        budget -= 10 * seg
prg, pos, alts, n_alts = snp_prg(ref, 60_000, 2)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
seeds = master_seeds(42, [n])
t0 = time.time()
want = oracle_map(prg, 10, list(reads), seeds, threads=256)
print(f"oracle: {n} reads in {time.time() - t0:.1f} s", flush=True)
qm = Quasimapper(Index(prg, 10))
qm.map_reads(np.ascontiguousarray(reads.reshape(-1)), flat_offsets(n, 150), seeds)
got = canonical_cov(qm.coverage())
assert got == want, "GPU differs from the oracle"
print(f"bit-exact: allele-sum, per-base, grouped counts and read counters of {n} reads; stats {got['stats']}; queues {qm.queue_counts()}")
