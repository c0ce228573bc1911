"""A PRG longer than 2^31 symbols (the range only unsigned 32-bit suffix-array indices reach; configs[4] = 3.46 G is
beyond what one test run affords): random reference of G bases + SNP sites, k = 14. Prints build phases, maps error-free
reads and checks the size-independent properties. Usage: python tools/scale_u32.py [G=2200000000] [N_SITES=1000000] [N_READS=200000]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from gramtools_amd import Index, Quasimapper, master_seeds  # noqa: E402
from gramtools_amd.synth import flat_offsets, random_ref, variant_sites, variant_prg, variant_haplotype, reads_from_haplotypes  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2_200_000_000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
t0 = time.time()
ref = random_ref(G, 1)
sites = variant_sites(ref, n_sites, 2, snp_frac=1.0)
prg, _ = variant_prg(ref, sites)
print(f"PRG {prg.size} symbols (2^31 = {2**31}), {n_sites} sites ({time.time() - t0:.0f} s)", flush=True)
assert prg.size > 2 ** 31
hap = variant_haplotype(ref, sites, 3)[0]
reads = reads_from_haplotypes([hap], n_reads, 150, 4)
del ref, hap
t0 = time.time()
ix = Index(prg, 14)
print(f"index: {time.time() - t0:.0f} s, {ix.info.index_bytes / 1e9:.1f} GB, k2 = {ix.info.kmer_size2}", flush=True)
sa_top = int(np.asarray(ix.sa()[:4096]).max())
print("suffix array entries above 2^31 among the first 4096:", int((np.asarray(ix.sa()[:4096]) >= 2 ** 31).sum()), "max", sa_top, flush=True)
seeds = master_seeds(42, [n_reads])
offs = flat_offsets(n_reads, 150)
qm = Quasimapper(ix)
t0 = time.time()
qm.map_reads(reads.reshape(-1), offs, seeds)
fwd = qm.coverage()
st = fwd.stats.as_dict()
print(f"mapped {n_reads} reads in {time.time() - t0:.2f} s: {st}", flush=True)
assert st["all"] == 2 * n_reads and st["exact_mapped"] >= n_reads
assert st["all"] == st["skipped"] + st["missing_kmer"] + st["no_extension"] + st["exact_mapped"]
rc = np.ascontiguousarray((5 - reads)[:, ::-1])
qm2 = Quasimapper(ix)
qm2.map_reads(rc.reshape(-1), offs, seeds)
back = qm2.coverage()
assert (fwd.raw_allele_sum == back.raw_allele_sum).all() and (fwd.raw_per_base == back.raw_per_base).all()
assert (fwd.raw_grouped == back.raw_grouped).all() and int(fwd.raw_allele_sum.sum()) > 0
print("properties hold beyond 2^31 symbols: every read maps, counter identity, strand symmetry", flush=True)
