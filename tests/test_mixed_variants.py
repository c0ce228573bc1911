"""Flat PRGs with SNPs, indels (empty alleles included), multi-allelic sites (dense grouped slots and the append
log) and adjacent sites; reads of ragged lengths 20..420 (shorter than one packed pair, longer than the
register-resident limit of 192 bases, block spans beyond the pack kernel's LDS window), k from 3 to 12 (LDS and
global-memory k-mer filter). Host emulation of the device headers (not gpu) and the HIP path (gpu) against the
oracle, bit-exact."""
import numpy as np
import pytest

from gramtools_amd import Index, Quasimapper
from gramtools_amd.synth import mixed_variant_prg, random_ref, simulate_haplotype_reads

from common import canonical_cov, flatten_reads, hostemu_map, oracle_map

KS = [7, 11, 5, 9, 12, 3]


def _case(seed, n_reads):
    ref = random_ref(20000, 3 + seed)
    prg, sites = mixed_variant_prg(ref, 250, 4 + seed)
    k = KS[seed % len(KS)]
    reads = [r for r in simulate_haplotype_reads(ref, sites, n_reads, 20, 420, 5 + seed) if len(r) >= k]
    rng = np.random.default_rng(seed)
    reads += [rng.integers(1, 5, size=int(rng.integers(k, 300))).astype(np.uint8) for _ in range(20)]  # unmappable
    bad = reads[0].copy()
    bad[len(bad) // 2] = 0  # a non-ACGT symbol: the read is skipped as a whole
    reads.append(bad)
    seeds = (np.arange(len(reads), dtype=np.uint64) * 7919 + seed).astype(np.uint32)
    return prg, k, reads, seeds


@pytest.mark.parametrize("seed", range(6))
def test_host_emulation_matches_oracle(seed):
    prg, k, reads, seeds = _case(seed, 300)
    want = oracle_map(prg, k, reads, seeds, rng_mode=seed % 2, threads=4)
    got, _, rc = hostemu_map(prg, k, reads, seeds, rng_mode=seed % 2)
    assert rc == 0
    assert got == want
    assert want["stats"]["exact_mapped"] >= 250 and want["stats"]["skipped"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_matches_oracle(seed):
    prg, k, reads, seeds = _case(seed, 3000)
    want = oracle_map(prg, k, reads, seeds, rng_mode=seed % 2, threads=8)
    qm = Quasimapper(Index(prg, k), rng_mode=seed % 2)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want


@pytest.mark.gpu
def test_gpu_long_reads_take_the_direct_pack_path():
    """1000-base reads: a pack block's 128 reads exceed the LDS window, the search reads them from memory."""
    ref = random_ref(30000, 77)
    prg, sites = mixed_variant_prg(ref, 300, 78)
    reads = simulate_haplotype_reads(ref, sites, 600, 900, 1000, 79)
    seeds = np.arange(len(reads), dtype=np.uint32) + 5
    want = oracle_map(prg, 9, reads, seeds, threads=8)
    qm = Quasimapper(Index(prg, 9))
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert want["stats"]["exact_mapped"] >= 500
