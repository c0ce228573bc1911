"""Experiment: P engines on ONE device, each with its own batches on a stream of its own (batches in flight side by side): what the
low-occupancy tail of a batch (nested PRGs: a few straggler tasks for 2 of the batch's 2.6 ms) costs when another batch can fill
the GPU meanwhile. Usage: python tools/exp/engines_in_flight.py CONFIG(1|2|3) P [N_READS per batch]"""
import sys
import time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import chr20_recipe, flat_offsets, pf3d7_recipe, random_ref, snp_prg, simulate_snp_reads_fast

which, P = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
if which == 2:
    prg, reads = pf3d7_recipe(23_300_000, 2000, 100_000, n * P, 22)
    k = 10
elif which == 3:
    prg, reads = chr20_recipe(64_444_167, 1_800_000, n * P, 32)
    k = 14
else:
    ref = random_ref(4411532, 1)
    prg, pos, alts, n_alts = snp_prg(ref, 60000, 2)
    reads = simulate_snp_reads_fast(ref, pos, alts, n_alts, n * P, 150, 1000)
    k = 10
ix = Index(prg, k)
seeds = master_seeds(42, [n * P])
offs = torch.from_numpy(flat_offsets(n, reads.shape[1]).astype(np.int64)).cuda()
eng = []
for p in range(P):
    r = np.ascontiguousarray(reads[p * n:(p + 1) * n]).reshape(-1)
    eng.append((Quasimapper(ix), torch.cuda.Stream(), torch.from_numpy(r).cuda(), torch.from_numpy(seeds[p * n:(p + 1) * n].astype(np.int64)).to(torch.int32).cuda()))


def loop(steps):
    for q, s, _, _ in eng:
        q.reset(stream=s.cuda_stream)
    for _ in range(steps):
        for q, s, d_r, d_s in eng:
            q.map_reads_device(d_r, offs, d_s, n, stream=s.cuda_stream)
    for q, _, _, _ in eng:
        q.sync()
    torch.cuda.synchronize()


loop(3)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    loop(10)
    best = min(best, (time.perf_counter() - t0) / 10)
print(f"configs[{which}], {P} engine(s) on one device, {n} reads per batch each: {best * 1e3:.3f} ms per round of {P} batches = {P * n / best / 1e6:.0f} M reads/s", flush=True)
