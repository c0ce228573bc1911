"""One configuration's index built and its reads mapped through the kernel-pipeline loop (bytes resident in HBM) and the
packed host feed (2-bit stream from page-locked memory), for use under rocprofv3 (tools/profile_round4.sh):
  python tools/profile_config.py 2|3|4|4s [N_READS=1000000] [STEPS=6]
    2   BASELINE configs[2]: 23.3 Mb + 2000 nested MSA regions + 100 k SNPs, k = 10
    3   BASELINE configs[3]: 64.4 Mb + 1.8 M sites, k = 14 (index 15.5 GB)
    4   BASELINE configs[4]: 3.1 Gb + 85 M sites, k = 14 (index 160 GB; 4-5 minutes of build)
    4s  configs[4] at an eighth of its length (400 Mb + 11 M sites)
    4k12  the same 400 Mb PRG with k = 12: 26 occurrences per k-mer, the seed cursor route of configs[4] in small"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds, pack_reads_2bit, PinnedArray  # noqa: E402
from gramtools_amd.synth import chr20_recipe, flat_offsets, genome_recipe_file  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = time.time()
if which == "2":
    from gramtools_amd.synth import pf3d7_recipe
    prg, reads = pf3d7_recipe(23_300_000, 2000, 100_000, n, 22)
    ix = Index(prg, 10)
elif which == "3":
    prg, reads = chr20_recipe(64_444_167, 1_800_000, n, 32)
    ix = Index(prg, 14)
else:
    G, S, seed = (3_100_000_000, 85_000_000, 61) if which == "4" else (400_000_000, 11_000_000, 51)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gmx_profile.prg")
    _, reads = genome_recipe_file(path, G, S, n, seed)
    ix = Index(path, 12 if which == "4k12" else 14)
    os.remove(path)
info = ix.info
print(f"configs[{which}]: {info.n_text - 1} symbols, {info.n_sites} sites, index {info.index_bytes / 1e9:.1f} GB, k2 = {info.kmer_size2}, "
      f"seed_shift {info.seed_shift} ({time.time() - t0:.0f} s)", flush=True)
seeds = master_seeds(42, [n])
offs = flat_offsets(n, reads.shape[1])
flat = np.ascontiguousarray(reads).reshape(-1)
qm = Quasimapper(ix)
d_r, d_o = torch.from_numpy(flat).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda()
d_s = torch.from_numpy(np.ascontiguousarray(seeds).view(np.int32).copy()).cuda()
stream = torch.cuda.current_stream().cuda_stream
for rep in range(2):  # (the first round sizes the workspace)
    qm.reset(stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        qm.map_reads_device(d_r, d_o, d_s, n, stream=stream)
    qm.sync()
    dt = (time.perf_counter() - t0) / steps
print(f"kernel pipeline: {dt * 1e3:.3f} ms per {n} reads = {n / dt / 1e6:.1f} M reads/s", flush=True)
print("queues of the last batch:", qm.queue_counts(), flush=True)
pk = pack_reads_2bit(flat, offs, uniform_len=reads.shape[1], pinned=True)
sd = PinnedArray(n, np.uint32)  # (round 5: page-locked seeds, read in place — pageable ones made every call register them and wait for
sd.array[:] = seeds             #  its own uploads, so that copies and kernels ran one after the other: round 4's packed-feed figures)
qm.seeds_in_place(True)
for rep in range(2):
    qm.reset()
    qm.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        qm.map_reads_packed(pk, sd.array, use_skip=False)
    qm.sync()
    dt = (time.perf_counter() - t0) / steps
print(f"packed host feed: {dt * 1e3:.3f} ms per {n} reads = {n / dt / 1e6:.1f} M reads/s", flush=True)
print("stats:", qm.coverage().stats.as_dict(), flush=True)
pk.close()
if os.environ.get("GMX_LIB", "").endswith("libgmx_stats.so"):  # debug build (-DGMX_LOOP_STATS): iteration mix of ONE batch
    import ctypes as C
    from gramtools_amd import _lib
    lib = _lib.load()
    lib.gmx_debug_loop_stats.restype = C.c_int
    lib.gmx_debug_loop_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 48)()
    lib.gmx_debug_loop_stats(buf, 1)
    qm.reset()
    qm.map_reads_device(d_r, d_o, d_s, n, stream=stream)
    qm.sync()
    lib.gmx_debug_loop_stats(buf, 1)
    names = ["fast iterations", "heavy TEXT", "heavy HIT", "heavy WIDE", "light only", "slow iterations", "lanes in heavy kinds",
             "lanes in slow iterations", "waves", "lanes in light kinds", "clk prologue", "clk loop", "clk epilogue", "live lanes (sum over iterations)"]
    for kk, kern in enumerate(["probe", "extend", "large-capacity"]):
        v = np.array(buf[kk * 16:kk * 16 + 14], dtype=np.float64)
        waves = max(v[8], 1)
        print(kern, f"waves={int(v[8])}")
        for nm, x in zip(names, v):
            print(f"    {nm:34s} {int(x):14d}   {x / waves:10.2f} per wave")
