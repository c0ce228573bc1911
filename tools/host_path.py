"""Host entry point (gmx_map_reads_host: the caller's pageable buffers, PCIe-inclusive): rate of the pipelined upload
against the serial one (GMX_HOST_SERIAL=1), and identical coverage. Usage: python tools/host_path.py [N_READS]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, simulate_snp_reads, snp_prg  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ref = random_ref(4_411_532, 1)
prg, pos, alts, n_alts = snp_prg(ref, 60_000, 2)
ix = Index(prg, 10)
reads = simulate_snp_reads(ref, pos, alts, n_alts, n_reads, 150, 1000)
flat = np.ascontiguousarray(reads.reshape(-1))
offs = flat_offsets(n_reads, 150)
seeds = master_seeds(42, [n_reads])
results = {}
for mode in ("serial", "pipelined", "serial", "pipelined"):
    if mode == "serial":
        os.environ["GMX_HOST_SERIAL"] = "1"
    else:
        os.environ.pop("GMX_HOST_SERIAL", None)
    qm = Quasimapper(ix)
    qm.map_reads(flat[: 150 * 1000], offs[:1001], seeds[:1000])  # warm-up: allocations
    qm.reset()
    t0 = time.perf_counter()
    qm.map_reads(flat, offs, seeds)
    dt = time.perf_counter() - t0
    cov = qm.coverage()
    results.setdefault(mode, cov)
    print(f"{mode:9s}: {n_reads} reads in {dt * 1e3:.1f} ms = {n_reads / dt / 1e6:.0f} M reads/s ({flat.nbytes / dt / 1e9:.1f} GB/s of read bytes); "
          f"stats {cov.stats.as_dict()}", flush=True)
a, b = results["serial"], results["pipelined"]
assert (a.raw_allele_sum == b.raw_allele_sum).all() and (a.raw_per_base == b.raw_per_base).all() and (a.raw_grouped == b.raw_grouped).all()
assert a.stats.as_dict() == b.stats.as_dict()
print("pipelined == serial coverage")
