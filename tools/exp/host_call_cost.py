"""Host-side cost of ONE gmx_map_reads_2bit_host call of 1 M reads with the device idle (no waiting: the launches themselves),
CPU seconds against wall seconds, and the same in a loop where the host runs ahead of the GPU (the slot waits)."""
import resource
import sys
import time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads_2bit, PinnedArray
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads_fast, snp_prg

ref = random_ref(4411532, 1)
prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
ix = Index(prg, 10)
n = 1_000_000
reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, n, 150, 1000)
sd = PinnedArray(n, np.uint32)
sd.array[:] = master_seeds(42, [n])
seeds = sd.array  # page-locked like the planes: pageable seeds would be registered and unregistered by every call
pk = pack_reads_2bit(np.ascontiguousarray(reads).reshape(-1), flat_offsets(n, 150), uniform_len=150, pinned=True)
qm = Quasimapper(ix)
qm.seeds_in_place(True)


def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


for _ in range(3):
    qm.map_reads_packed(pk, seeds, use_skip=False)
qm.sync()
walls = []
for _ in range(20):
    qm.sync()
    time.sleep(0.002)
    c0, t0 = cpu(), time.perf_counter()
    qm.map_reads_packed(pk, seeds, use_skip=False)
    walls.append((time.perf_counter() - t0, cpu() - c0))
print("one call, device idle: wall %.0f us, CPU %.0f us (median of 20)" % (np.median([w for w, _ in walls]) * 1e6, np.median([c for _, c in walls]) * 1e6))
for steps in (50, 400):
    qm.sync()
    c0, t0 = cpu(), time.perf_counter()
    for _ in range(steps):
        qm.map_reads_packed(pk, seeds, use_skip=False)
    t1, c1 = time.perf_counter(), cpu()
    qm.sync()
    t2, c2 = time.perf_counter(), cpu()
    print(f"{steps} calls back to back: {1e3 * (t1 - t0) / steps:.3f} ms wall per call, CPU {1e3 * (c1 - c0) / steps:.3f} ms per call; final sync wall {1e3 * (t2 - t1):.2f} ms CPU {1e3 * (c2 - c1):.2f} ms")

# which threads burn the CPU while the host waits for the device?
import os


def thread_times():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out


for mode in ("default", ):
    qm.sync()
    a = thread_times()
    t0 = time.perf_counter()
    for _ in range(1500):
        qm.map_reads_packed(pk, seeds, use_skip=False)
    qm.sync()
    wall = time.perf_counter() - t0
    b = thread_times()
    print(f"1500 calls: wall {wall:.2f} s; threads with CPU time:")
    for tid, (comm, t) in sorted(b.items(), key=lambda kv: -(kv[1][1] - a.get(kv[0], ("", 0))[1])):
        d = t - a.get(tid, ("", 0))[1]
        if d > 0.01:
            print(f"   tid {tid} {comm:20s} {d:.2f} s{'  (the calling thread)' if tid == os.getpid() else ''}")
