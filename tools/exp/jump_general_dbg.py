"""gmx_cover_jump called from the general coverage instances (exp builds -DGMX_EXP_DEVICE_JUMP): 37 loci per read, GPU first or oracle first"""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
order = sys.argv[1]
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import flat_offsets, random_ref, snp_prg, simulate_snp_reads
ref = random_ref(30000, 77); prg, pos, alts, n_alts = snp_prg(ref, 7400, 78, multi_allelic_frac=0.1)
reads = simulate_snp_reads(ref, pos, alts, n_alts, 600, 150, 79)
seeds = master_seeds(13, [600]); offs = flat_offsets(600, 150)
def gpu():
    qm = Quasimapper(Index(prg, 7)); qm.map_reads(reads.reshape(-1), offs, seeds); c = qm.coverage()
    return int(c.raw_allele_sum.sum()), int(c.raw_grouped.sum()), int(c.raw_per_base.sum())
def cpu():
    from common import oracle_map
    w = oracle_map(prg, 7, list(reads), seeds, threads=8)
    return sum(sum(x) for s in w['allele_sum'] for x in ([s] if isinstance(s, list) and s and isinstance(s[0], int) else s)) if False else w['depth']
if order == "gpu_first":
    g = gpu(); print("gpu", g); print("oracle depth", cpu()); print("gpu again", gpu())
else:
    print("oracle depth", cpu()); print("gpu", gpu()); print("gpu again", gpu())
