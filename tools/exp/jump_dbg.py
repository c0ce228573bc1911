"""Experiment (round 5): counters inside gmx_cover_jump for the 37-loci workload of test_dense_sites_matches_oracle[7400];
GMX_LIB = a build with -DGMX_JUMP_DBG. [0..4]: reads that start inside an allele — calls, check pass refused, check pass
accepted, operations it counted, operations the recording pass made; [8..12]: the same for the other reads."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from common import oracle_map, canonical_cov
from gramtools_amd import Index, Quasimapper, master_seeds, _lib
from gramtools_amd.synth import random_ref, snp_prg, simulate_snp_reads, flat_offsets
ref = random_ref(30000, 40 + 7400)
prg, pos, alts, n_alts = snp_prg(ref, 7400, 40 + 7400 + 1, multi_allelic_frac=0.1)
reads = simulate_snp_reads(ref, pos, alts, n_alts, 3000, 150, 40 + 7400 + 2)
seeds = master_seeds(13, [3000])
want = oracle_map(prg, 7, list(reads), seeds, threads=8)
qm = Quasimapper(Index(prg, 7))
qm.map_reads(reads.reshape(-1), flat_offsets(3000, 150), seeds)
got = canonical_cov(qm.coverage())
out = (C.c_ulonglong * 16)()
lib = _lib.load()
lib.gmx_debug_jump_dbg(out)
print("equal to the oracle:", got == want, "| increments", sum(sum(x) for x in got["allele_sum"]), "oracle", sum(sum(x) for x in want["allele_sum"]))
print("all 16:", list(out))
print("other reads   :", list(out[8:13]))
