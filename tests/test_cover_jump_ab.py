"""Device-side A/B guard of the geometry-record coverage (gmx_cover_jump: gmx_cover_jump_kernel for compact records, and —
since round 5 — the general coverage instances for paths of 17+ loci) against the graph walk (GMX_NO_COVER_JUMP=1: no kernel
sees the geometry records, coverage/allele_base.cpp:137-296 as the reference walks it). Round 4 met a wrong-result build of
this routine (root-caused in round 5: profiles/round5/jump_ptr_form_*); this test would have caught it on the
device without the oracle in the loop: reads that start inside alleles, paths of 7 to 37 loci, long and empty alleles,
adjacent sites, both strands. The raw accumulators of the two engines must be identical."""
import numpy as np
import pytest

from common import canonical_cov, flatten_reads, oracle_map
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import (flat_offsets, mixed_variant_prg, random_ref, simulate_haplotype_reads, simulate_snp_reads,
                                 snp_prg)

pytestmark = pytest.mark.gpu


def _both(monkeypatch, ix, flat, offs, seeds):
    covs = []
    for walk in (False, True):
        if walk:
            monkeypatch.setenv("GMX_NO_COVER_JUMP", "1")
        else:
            monkeypatch.delenv("GMX_NO_COVER_JUMP", raising=False)
        qm = Quasimapper(ix)
        qm.map_reads(flat, offs, seeds)
        covs.append(qm.coverage())
    monkeypatch.delenv("GMX_NO_COVER_JUMP", raising=False)
    a, b = covs
    assert (a.raw_allele_sum == b.raw_allele_sum).all()
    assert (a.raw_per_base == b.raw_per_base).all()
    assert (a.raw_grouped == b.raw_grouped).all()
    assert a.stats.as_dict() == b.stats.as_dict()
    return a


@pytest.mark.parametrize("n_sites", [1500, 7400])
def test_jump_equals_walk_on_dense_snps(monkeypatch, n_sites):
    """7 and 37 loci per read: compact records (the lean kernel) and the general instances' paths"""
    ref = random_ref(30000, 900 + n_sites)
    prg, pos, alts, n_alts = snp_prg(ref, n_sites, 901 + n_sites, multi_allelic_frac=0.1)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, 20000, 150, 902 + n_sites)
    seeds = master_seeds(7, [reads.shape[0]])
    ix = Index(prg, 7)
    assert ix.info.n_jump_sites > 0.9 * ix.n_sites
    cov = _both(monkeypatch, ix, reads.reshape(-1), flat_offsets(*reads.shape), seeds)
    assert cov.stats.exact_mapped_reads_count >= reads.shape[0]


@pytest.mark.parametrize("seed,density,max_len", [(1, 6, 30), (2, 25, 12), (3, 60, 30), (4, 10, 3)])
def test_jump_equals_walk_and_oracle_on_mixed_sites(monkeypatch, seed, density, max_len):
    """alleles of 0-30 bases (most reads start or end inside one), 2-9 alleles, adjacent sites, ragged reads"""
    G = 30000
    ref = random_ref(G, 7000 + seed)
    prg, st = mixed_variant_prg(ref, G // density, 7100 + seed, max_alleles=5 + seed, max_len=max_len, adjacent_prob=0.1)
    reads = [r for r in simulate_haplotype_reads(ref, st, 6000, 40, 200, 7200 + seed) if len(r) >= 9]
    seeds = (np.arange(len(reads), dtype=np.uint64) * 2654435761 + seed).astype(np.uint32)
    flat, offs = flatten_reads(reads)
    cov = _both(monkeypatch, Index(prg, 9), flat, offs, seeds)
    assert canonical_cov(cov) == oracle_map(prg, 9, reads, seeds, threads=8)
