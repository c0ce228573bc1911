"""Iteration mix of the wave loop (debug build with -DGMX_LOOP_STATS; see tools/loop_stats.sh)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from gramtools_amd import Index, Quasimapper, _lib  # noqa: E402

from gramtools_amd import master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ref = random_ref(bench.GENOME, 1)
if len(sys.argv) > 2:  # fraction of the genome replaced by 10 copies each of 1-5 kb segments
    rng = np.random.default_rng(5)
    budget = int(ref.size * float(sys.argv[2]))
    while budget > 0:
        seg = int(rng.integers(1000, 5001))
        src = int(rng.integers(0, ref.size - seg))
        piece = ref[src:src + seg].copy()
        for _ in range(10):
            dst = int(rng.integers(0, ref.size - seg))
            ref[dst:dst + seg] = piece
        budget -= 10 * seg
prg, pos, alts, n_alts = snp_prg(ref, bench.N_SITES, 2)
ix = Index(prg, bench.KMER)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, bench.READ_LEN, 1000)
seeds = master_seeds(42, [n])
offsets = flat_offsets(n, bench.READ_LEN)
qm = Quasimapper(ix, device=0)
lib = _lib.load()
lib.gmx_debug_loop_stats.restype = C.c_int
lib.gmx_debug_loop_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 48)()
qm.map_reads(reads.reshape(-1), offsets, seeds)
qm.sync()
print("queues:", qm.queue_counts())
lib.gmx_debug_loop_stats(buf, 1)
names = ["fast iterations", "heavy TEXT", "heavy HIT", "heavy WIDE", "light only", "slow iterations",
         "lanes in heavy kinds", "lanes in slow iterations", "waves", "lanes in light kinds",
         "clk prologue", "clk loop", "clk epilogue", "live lanes (sum over iterations)"]
for k, kern in enumerate(["probe", "extend", "large-capacity"]):
    v = np.array(buf[k * 16:k * 16 + 14], dtype=np.float64)
    waves = max(v[8], 1)
    print(kern, f"waves={int(v[8])}")
    for n, x in zip(names, v):
        print(f"    {n:28s} {int(x):12d}   {x / waves:8.2f} per wave")
