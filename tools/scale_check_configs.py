"""BASELINE.json configs[2] and configs[3] by their full recipes (gramtools_amd/synth.py): device-resident
rate of the bench's loop and the queues of the last batch. Usage: python tools/scale_check_configs.py 2|3 [N_READS]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import chr20_recipe, flat_offsets, pf3d7_recipe  # noqa: E402

which = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
t0 = time.time()
if which == 2:
    prg, reads = pf3d7_recipe(23_300_000, 2000, 100_000, n, 22)
    k = 10
else:
    prg, reads = chr20_recipe(64_444_167, 1_800_000, n, 32)
    k = 14
print(f"configs[{which}]: PRG {prg.size} symbols, {reads.shape[0]} reads ({time.time() - t0:.1f} s)", flush=True)
t0 = time.time()
ix = Index(prg, k)
print(f"index: {time.time() - t0:.1f} s, {ix.info.index_bytes / 1e9:.2f} GB, k2 = {ix.info.kmer_size2}, nested = {ix.info.is_nested}", flush=True)
n = reads.shape[0]
seeds = master_seeds(42, [n])
offs = flat_offsets(n, reads.shape[1])
d_reads = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
qm = Quasimapper(ix)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    qm.reset(stream=stream)
    qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
qm.sync()
steps = 10
t0 = time.perf_counter()
for _ in range(steps):
    qm.reset(stream=stream)
    qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
qm.sync()
dt = (time.perf_counter() - t0) / steps
st = qm.coverage().stats.as_dict()
print(f"device-resident: {dt * 1e3:.2f} ms per {n} reads = {n / dt / 1e6:.0f} M reads/s; stats {st}", flush=True)
print("queues of the last batch:", qm.queue_counts(), flush=True)
