"""configs[2] recipe in medium size: random sequence with nested "MSA regions" (depth <= 3, empty alleles, adjacent
sites), reads from pre-drawn haplotypes; bit-exact against the oracle on a sample, device-resident rate on all reads.
Usage: python tools/scale_check_nested.py N_REGIONS SPACER K N_READS N_ORACLE"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import nested_regions_prg, simulate_graph_reads  # noqa: E402
from common import canonical_cov, flatten_reads, oracle_map  # noqa: E402

n_regions, spacer, k, n_reads, n_oracle = (int(x) for x in sys.argv[1:6])
t0 = time.time()
prg = nested_regions_prg(n_regions, 7, spacer_lo=spacer // 2, spacer_hi=spacer * 3 // 2)
print(f"PRG {len(prg)} symbols, {n_regions} nested regions ({time.time() - t0:.1f} s)", flush=True)
t0 = time.time()
reads = simulate_graph_reads(prg, n_reads, 150, 11, n_haps=6)
print(f"{len(reads)} reads ({time.time() - t0:.1f} s)", flush=True)
t0 = time.time()
ix = Index(prg, k)
print(f"index: {time.time() - t0:.1f} s, {ix.info.index_bytes / 1e6:.0f} MB, nested={ix.info.is_nested}", flush=True)
seeds = master_seeds(42, [len(reads)])
flat, offs = flatten_reads(reads)
sample = reads[:n_oracle]
t0 = time.time()
want = oracle_map(prg, k, sample, seeds[:n_oracle], threads=64)
print(f"oracle: {n_oracle} reads in {time.time() - t0:.1f} s", flush=True)
qm = Quasimapper(ix)
sflat, soffs = flatten_reads(sample)
qm.map_reads(sflat, soffs, seeds[:n_oracle])
assert canonical_cov(qm.coverage()) == want, "GPU differs from the oracle"
print("bit-exact against the oracle; queues:", qm.queue_counts(), flush=True)
d_reads = torch.from_numpy(flat).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).cuda()
qm2 = Quasimapper(ix)
for _ in range(2):
    qm2.reset()
    qm2.map_reads_device(d_reads, d_offs, d_seeds, len(reads))
qm2.sync()
t0 = time.perf_counter()
for _ in range(5):
    qm2.reset()
    qm2.map_reads_device(d_reads, d_offs, d_seeds, len(reads))
qm2.sync()
dt = (time.perf_counter() - t0) / 5
print(f"device-resident: {dt * 1e3:.2f} ms per {len(reads)} reads = {len(reads) / dt / 1e6:.0f} M reads/s; "
      f"stats {qm2.coverage().stats.as_dict()}; queues {qm2.queue_counts()}", flush=True)
