"""Experiment: the bench job (1 M reads) split over P engines on the same device, each on its own stream, joined at
the end of every step. Estimates what batches in flight side by side would buy. Usage: python tools/two_engines.py P [repeats]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

P = int(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
G, n_sites, k, n = 4411532, 60000, 10, 1 << 20
ref = random_ref(G, 1)
if frac > 0:
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
prg, pos, alts, n_alts = snp_prg(ref, n_sites, 2)
ix = Index(prg, k)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
seeds = master_seeds(42, [n])
part = n // P
engines = []
for p in range(P):
    r = reads[p * part:(p + 1) * part]
    engines.append((Quasimapper(ix), torch.cuda.Stream(), torch.cuda.Event(),
                    torch.from_numpy(np.ascontiguousarray(r).reshape(-1)).cuda(),
                    torch.from_numpy(flat_offsets(part, 150).astype(np.int64)).cuda(),
                    torch.from_numpy(seeds[p * part:(p + 1) * part].astype(np.int64)).to(torch.int32).cuda()))


def step():
    for qm, st, ev, dr, do, ds in engines:
        qm.reset(stream=st.cuda_stream)
        qm.map_reads_device(dr, do, ds, part, stream=st.cuda_stream)
        ev.record(st)
    for _, st, _, _, _, _ in engines:  # join: a step is a whole job
        for _, _, ev, _, _, _ in engines:
            st.wait_event(ev)


for _ in range(3):
    step()
torch.cuda.synchronize()
steps = 30
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"P={P} repeats={frac}: {dt * 1e6:.0f} us per step of {part * P} reads = {part * P / dt / 1e6:.0f} M reads/s", flush=True)
