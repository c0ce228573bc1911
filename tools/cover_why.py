"""Why tasks leave a coverage instance for the next one, and where the instances' time goes (debug build with
-DGMX_LOOP_STATS), on the configs[2] recipe.  gpurun -- 'bash tools/cover_why.sh'"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, _lib, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, pf3d7_recipe  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prg, reads = pf3d7_recipe(23_300_000, 2000, 100_000, n, 22)
ix = Index(prg, 10)
n = reads.shape[0]
seeds = master_seeds(42, [n])
offs = flat_offsets(n, reads.shape[1])
qm = Quasimapper(ix)
lib = _lib.load()
for name, size in (("gmx_debug_cover_why", 32), ("gmx_debug_cover_stats", 96), ("gmx_debug_coop_stats", 48)):
    getattr(lib, name).restype = C.c_int
    getattr(lib, name).argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
why = (C.c_ulonglong * 32)()
stats = (C.c_ulonglong * 96)()
cstats = (C.c_ulonglong * 48)()
flat = np.ascontiguousarray(reads).reshape(-1)
qm.map_reads(flat, offs, seeds)
qm.sync()
lib.gmx_debug_cover_why(why, 1)
lib.gmx_debug_cover_stats(stats, 1)
lib.gmx_debug_coop_stats(cstats, 1)
qm.reset()
qm.map_reads(flat, offs, seeds)
qm.sync()
print("queues:", qm.queue_counts())
lib.gmx_debug_cover_why(why, 1)
lib.gmx_debug_cover_stats(stats, 1)
lib.gmx_debug_coop_stats(cstats, 1)
names = {0: "one lane, regular scratch", 1: "one lane, large scratch (global memory)", 2: "one lane, large-capacity tasks",
         3: "one lane, first scratch", 4: "one lane, large-capacity tasks, first part", 5: "one lane, instance tasks",
         6: "cooperative: item scratch", 7: "cooperative: class scratch"}
for lst in range(8):
    v = [int(why[lst * 4 + k]) for k in range(4)]
    if sum(v):
        print(f"instance {lst} [{names[lst]}]: exceeded loci {v[0]}, key sites {v[1]}, hull {v[2]}, items/units {v[3]}")
ph = ["items", "loci+keys", "sort+draw", "class loci+hull", "(unused)", "record", "before the task", "tasks"]
for lst in range(6):
    v = np.array(stats[lst * 16:lst * 16 + 16], dtype=np.float64)
    if v[7] == 0:
        continue
    print(f"instance {lst}: {int(v[7])} tasks; mean us per task: " + ", ".join(f"{ph[k]} {v[k] / v[7] / 100:.1f}" for k in (0, 1, 2, 3, 5)))
coop = ["units", "loci + keys", "classes + draw", "class merge + record"]
print("cooperative instances (wave-level wall time, us per round of 4 tasks):")
for lst in (2, 3, 5):
    v = np.array(cstats[lst * 8:lst * 8 + 8], dtype=np.float64)
    if v[7]:
        print(f"  instance {lst}: {int(v[7])} rounds; " + ", ".join(f"{coop[k]} {v[k] / v[7] / 100:.1f}" for k in range(4)))
