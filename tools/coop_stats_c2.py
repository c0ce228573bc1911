"""configs[2] (nested MSA regions): wave-level wall time of the cooperative coverage instances' phases and the serial
instances' phases, from a -DGMX_LOOP_STATS build (GMX_LIB=.../libgmx_stats.so). Usage: python tools/coop_stats_c2.py [N_READS] [3]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, _lib, master_seeds  # noqa: E402
from gramtools_amd.synth import chr20_recipe, flat_offsets, pf3d7_recipe  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
if len(sys.argv) > 2 and sys.argv[2] == "3":  # configs[3] instead: flat, SNP / indel mix
    prg, reads = chr20_recipe(64_444_167, 1_800_000, n, 32)
    ix = Index(prg, 14)
else:
    prg, reads = pf3d7_recipe(23_300_000, 2000, 100_000, n, 22)
    ix = Index(prg, 10)
seeds = master_seeds(42, [n])
offs = flat_offsets(n, reads.shape[1])
qm = Quasimapper(ix)
lib = _lib.load()
flat = np.ascontiguousarray(reads).reshape(-1)
qm.map_reads(flat, offs, seeds)
qm.sync()
coop = (C.c_ulonglong * 48)()
cov = (C.c_ulonglong * 96)()
lib.gmx_debug_coop_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.gmx_debug_cover_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.gmx_debug_coop_stats(coop, 1)
lib.gmx_debug_cover_stats(cov, 1)
qm.reset()
qm.map_reads(flat, offs, seeds)
qm.sync()
print("queues:", qm.queue_counts())
lib.gmx_debug_coop_stats(coop, 1)
lib.gmx_debug_cover_stats(cov, 1)
for lst in range(6):
    v = np.array(coop[lst * 8:lst * 8 + 8], dtype=np.float64)
    if v[7] == 0:
        continue
    print(f"coop<{lst}>: {int(v[7])} rounds (4 tasks each); per round, 10 ns units -> us: units {v[0] / v[7] / 100:.1f}, loci+keys {v[1] / v[7] / 100:.1f}, "
          f"classes+draw {v[2] / v[7] / 100:.1f}, class merge+record {v[3] / v[7] / 100:.1f}")
names = ["items", "loci+keys", "sort+draw", "class loci+hull", "(unused)", "record", "before the task", "tasks"]
for lst in range(6):
    v = np.array(cov[lst * 16:lst * 16 + 16], dtype=np.float64)
    if v[7] == 0:
        continue
    print(f"serial LIST {lst}: {int(v[7])} tasks: " + ", ".join(f"{names[k]} {v[k] / v[7] / 100:.1f} us" for k in (0, 1, 2, 3, 5)))
