"""A fixed-seed slice of tools/fuzz_parity.py under pytest: random FLAT PRGs — a site per 4-150 bases, alleles of 0-30 bases,
2-9 alleles (dense group counters and the append log), adjacent sites, ragged reads on both strands — and random NESTED
ones, the HIP path against the oracle, bit-exact. (The tool itself runs open-ended; 714 cases: profiles/round4/fuzz_parity.txt.)"""
import numpy as np
import pytest

from common import canonical_cov, flatten_reads, oracle_map
from gramtools_amd import Index, Quasimapper
from gramtools_amd.synth import (bracket_to_ints, mixed_variant_prg, nested_prg, random_ref, simulate_graph_reads,
                                 simulate_haplotype_reads)

pytestmark = pytest.mark.gpu


def flat_case(seed):
    """the case generator of tools/fuzz_parity.py, 1000 reads instead of 2500"""
    rng = np.random.default_rng(1000 + seed)
    G = int(rng.integers(5_000, 40_000))
    density = [4, 6, 10, 25, 60, 150][int(rng.integers(0, 6))]  # bases per site
    max_len = [1, 3, 6, 12, 30][int(rng.integers(0, 5))]
    ref = random_ref(G, 7 * seed + 1)
    prg, st = mixed_variant_prg(ref, max(G // density, 3), 7 * seed + 2, max_alleles=int(rng.integers(3, 10)), max_len=max_len,
                                adjacent_prob=float(rng.choice([0.0, 0.05, 0.3])))
    k = int(rng.choice([5, 7, 9, 11]))
    lo = int(rng.integers(k, 120))
    reads = [r for r in simulate_haplotype_reads(ref, st, 1000, lo, lo + int(rng.integers(1, 250)), 7 * seed + 3) if len(r) >= k]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 2654435761 + seed).astype(np.uint32)
    return prg, k, reads, seeds


@pytest.mark.parametrize("seed", range(5000, 5040))
def test_fuzz_flat(seed):
    prg, k, reads, seeds = flat_case(seed)
    mode = seed % 2
    want = oracle_map(prg, k, reads, seeds, rng_mode=mode, threads=8)
    qm = Quasimapper(Index(prg, k), rng_mode=mode)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want


@pytest.mark.parametrize("seed", range(6000, 6020))
def test_fuzz_nested(seed):
    """nested bracket PRGs (depth <= 4, empty alleles, adjacent sites) of a few hundred symbols, reads of 8-60 bases"""
    rng = np.random.default_rng(seed)
    s = nested_prg(seed, n_top=int(rng.integers(3, 12)), max_depth=int(rng.integers(1, 5)), seq_max=int(rng.integers(2, 12)))
    prg = bracket_to_ints(s)
    L, k = int(rng.integers(8, 60)), int(rng.integers(2, 7))
    reads = [r for r in simulate_graph_reads(prg, 400, L, seed + 1) if len(r) >= k]
    reads += [rng.integers(1, 5, size=L).astype(np.uint8) for _ in range(20)]
    seeds = rng.integers(0, 2 ** 32, size=len(reads), dtype=np.uint64).astype(np.uint32)
    mode = seed % 2
    want = oracle_map(prg, k, reads, seeds, rng_mode=mode, threads=8)
    qm = Quasimapper(Index(prg, k), rng_mode=mode)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
