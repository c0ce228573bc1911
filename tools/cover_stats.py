"""Wall time of the general coverage routine's phases per coverage instance (debug build with -DGMX_LOOP_STATS).
  gpurun -- 'bash tools/cover_stats.sh 0.002'"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from gramtools_amd import Index, Quasimapper, _lib, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
n = 1_000_000
ref = random_ref(bench.GENOME, 1)
rng = np.random.default_rng(5)
budget = int(ref.size * frac)
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
lib.gmx_debug_cover_stats.restype = C.c_int
lib.gmx_debug_cover_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 96)()
qm.map_reads(reads.reshape(-1), offsets, seeds)
qm.sync()
lib.gmx_debug_cover_stats(buf, 1)
qm.reset()
qm.map_reads(reads.reshape(-1), offsets, seeds)
qm.sync()
print("queues:", qm.queue_counts())
lib.gmx_debug_cover_stats(buf, 1)
names = ["items", "loci+keys", "sort+draw", "class loci+hull", "(unused)", "record", "before the task", "tasks"]
for lst in range(6):
    v = np.array(buf[lst * 16:lst * 16 + 16], dtype=np.float64)
    tasks = v[7]
    if tasks == 0:
        continue
    print(f"LIST {lst}: {int(tasks)} tasks")
    for k in (6, 0, 1, 2, 3, 5):
        print(f"    {names[k]:18s} mean {v[k] / tasks / 100:8.2f} us   max {v[8 + k] / 100:8.2f} us")
