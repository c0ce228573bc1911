"""Experiment: what read locality is worth to the kernels. The bench workload with its reads in random order, sorted by
region (8 regions of the genome, random within), and fully sorted by start position; device-resident rate + kernel times.
Usage: python tools/sorted_reads_exp.py"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
import bench  # noqa: E402
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

n = 1 << 20
ref = random_ref(bench.GENOME, 1)
prg, pos, alts, n_alts = snp_prg(ref, bench.N_SITES, 2)
ix = Index(prg, bench.KMER)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n, 150, 1000)
starts = np.random.default_rng(1000).integers(0, ref.size - 150 + 1, size=n)  # the simulator's first draw
seeds = master_seeds(42, [n])
offs = flat_offsets(n, 150)
orders = {"random": np.arange(n), "by region": np.argsort(starts // (ref.size // 8 + 1), kind="stable"), "sorted": np.argsort(starts)}
for name, order in orders.items():
    r = np.ascontiguousarray(reads[order])
    d_reads = torch.from_numpy(r.reshape(-1)).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_seeds = torch.from_numpy(seeds[order].astype(np.int64)).to(torch.int32).cuda()
    qm = Quasimapper(ix)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        qm.reset(stream=stream)
        qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
    qm.sync()
    qm.enable_timing(True)
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        qm.reset(stream=stream)
        qm.map_reads_device(d_reads, d_offs, d_seeds, n, stream=stream)
    qm.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = qm.timing()
    print(f"{name:10s}: {dt * 1e6:6.0f} us per step = {n / dt / 1e6:6.0f} M reads/s; extend kernel {tm['search_ms'] / max(tm['search_launches'], 1) * 1e3:6.1f} us, "
          f"after it {tm['cover_ms'] / max(tm['cover_launches'], 1) * 1e3:6.1f} us", flush=True)
    qm.close()
