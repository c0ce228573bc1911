"""Nested PRGs at a size where the text-form steps, the general jump programs, parking and the tiers all take part:
random sequence interleaved with 60-150 nested bracket regions (depth <= 3, empty alleles, adjacent sites), 100-200 bp
reads from pre-drawn haplotypes, both orientations. Host emulation (not gpu) and the HIP path (gpu) against the
oracle, bit-exact. Single-instance tasks take gmx_cover_single_nested (gmx_cover.h), which hands tasks with more loci
than it holds to the general routine: run with capacities 0-2 as well, so that both routes are pinned."""
import numpy as np
import pytest

from gramtools_amd import Index, Quasimapper
from gramtools_amd.synth import nested_regions_prg, simulate_graph_reads

from common import canonical_cov, flatten_reads, hostemu_map, oracle_map


def _case(seed, n_regions, n_reads):
    prg = nested_regions_prg(n_regions, seed)
    rng = np.random.default_rng(seed)
    L = int(rng.integers(100, 201))
    reads = simulate_graph_reads(prg, n_reads, L, seed + 1, n_haps=12)
    k = int(rng.integers(5, 11))
    reads = [r for r in reads if len(r) >= k]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 104729 + seed).astype(np.uint32)
    return prg, k, reads, seeds


@pytest.mark.parametrize("seed", range(4))
def test_host_emulation_matches_oracle(seed):
    prg, k, reads, seeds = _case(seed, 60, 300)
    want = oracle_map(prg, k, reads, seeds, rng_mode=seed % 2, threads=4)
    got, _, rc = hostemu_map(prg, k, reads, seeds, rng_mode=seed % 2)
    assert rc == 0
    assert got == want
    assert want["stats"]["exact_mapped"] >= 250


@pytest.mark.parametrize("single_loci", [0, 1, 2])
def test_single_instance_routine_hands_over_to_the_general_one(single_loci):
    prg, k, reads, seeds = _case(2, 60, 300)
    want = oracle_map(prg, k, reads, seeds, threads=4)
    got, _, rc = hostemu_map(prg, k, reads, seeds, single_loci=single_loci)
    assert rc == 0
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_gpu_matches_oracle(seed):
    prg, k, reads, seeds = _case(10 + seed, 150, 3000)
    want = oracle_map(prg, k, reads, seeds, rng_mode=seed % 2, threads=8)
    qm = Quasimapper(Index(prg, k), rng_mode=seed % 2)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert want["stats"]["exact_mapped"] >= 2500


def _repeated_nested_prg(seed, copies):
    """The same block (spacer + nested region + spacer) pasted `copies` times between random spacers: a read inside
    it has that many mapping instances, each crossing nested sites of its own copy."""
    from gramtools_amd.synth import bracket_to_ints, nested_prg
    rng = np.random.default_rng(seed)
    letters = "acgt"

    def spacer(n):
        return "".join(letters[int(x)] for x in rng.integers(0, 4, size=n))

    block = spacer(90) + nested_prg(seed * 31 + 1, n_top=2, max_depth=3, seq_max=5) + spacer(90)
    parts = []
    for _ in range(copies):
        parts.append(spacer(int(rng.integers(150, 400))))
        parts.append(block)
    parts.append(spacer(300))
    return bracket_to_ints("".join(parts))


@pytest.mark.parametrize("seed,copies", [(1, 3), (2, 8)])
def test_host_emulation_matches_oracle_on_repeated_nested_blocks(seed, copies):
    prg = _repeated_nested_prg(seed, copies)
    reads = [r for r in simulate_graph_reads(prg, 400, 70, seed + 5, n_haps=8) if len(r) >= 7]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 7919 + seed).astype(np.uint32)
    want = oracle_map(prg, 7, reads, seeds, threads=4)
    got, _, rc = hostemu_map(prg, 7, reads, seeds)
    assert rc == 0
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed,copies", [(3, 4), (4, 9), (5, 20), (6, 70)])
def test_gpu_matches_oracle_on_repeated_nested_blocks(seed, copies):
    """Multi-instance tasks whose items carry nested loci: instance lanes, the cooperative coverage instance (up to 16
    items), the one-lane instances and the split search behind them."""
    prg = _repeated_nested_prg(seed, copies)
    reads = [r for r in simulate_graph_reads(prg, 4000, 80, seed + 5, n_haps=8) if len(r) >= 8]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 7919 + seed).astype(np.uint32)
    want = oracle_map(prg, 8, reads, seeds, rng_mode=seed % 2, threads=8)
    qm = Quasimapper(Index(prg, 8), rng_mode=seed % 2)
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want
    assert want["stats"]["exact_mapped"] >= 2000
