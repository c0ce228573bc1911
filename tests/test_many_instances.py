"""Reads with thousands of mapping instances, and the tiers they pass through.

The reference has no capacity anywhere on the path: handle_allele_encapsulated_states iterates every position of every
final interval (encapsulated_search.cpp:30-107) and the instance selection builds its classes in a std::map
(coverage_common.cpp:85-146). The engine has fixed pools per tier and, behind them, a heap-backed last tier
(gmx_engine.hip: gmx_tail_stage), so that no valid read makes `gram genotype` fail. PRG: N tandem copies of one unit
holding a SNP site (every copy its own site): a read maps once per copy — N mapping instances, N equivalence classes
(one per copy's site), or N non-variant positions when the read misses the site.
"""
import numpy as np
import pytest

from common import oracle_map, hostemu_map, canonical_cov, flatten_reads
from gramtools_amd import Index, Quasimapper, master_seeds, GmxError


def tandem_prg(n_copies, unit_len, seed, flank=40):
    rng = np.random.default_rng(seed)
    unit = rng.integers(1, 5, size=unit_len, dtype=np.uint8)
    site_at = unit_len // 2
    alt = np.uint8(unit[site_at] % 4 + 1)
    left, right = rng.integers(1, 5, size=flank, dtype=np.uint8), rng.integers(1, 5, size=flank, dtype=np.uint8)
    out = [int(x) for x in left]
    for c in range(n_copies):
        m = 5 + 2 * c
        out += [int(x) for x in unit[:site_at]] + [m, int(unit[site_at]), m + 1, int(alt), m + 1] + [int(x) for x in unit[site_at + 1:]]
    out += [int(x) for x in right]
    return np.asarray(out, dtype=np.uint32), unit, site_at, alt


def tandem_reads(unit, site_at, alt, read_len, seed, n=6):
    """Reads from inside the tandem array (offsets into the unit; ref or alt allele at the site), both strands."""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n):
        u = unit.copy()
        if i % 2:
            u[site_at] = alt
        rep = np.tile(u, read_len // unit.size + 3)
        st = int(rng.integers(0, unit.size))
        r = rep[st:st + read_len].copy()
        if i % 3 == 2:
            r = (5 - r)[::-1].copy()
        reads.append(r)
    reads.append(unit[:site_at - 1][-min(site_at - 1, read_len):].copy())  # misses the site: non-variant instances only
    return [r for r in reads if len(r) >= 8]


def test_hostemu_many_instances_match_oracle():
    """1 500 copies through the device headers on the host: beyond every fixed scratch (1 024 items), so the
    task's selection runs in the dynamically sized scratch (CoverEnvDyn's mirror)."""
    prg, unit, site_at, alt = tandem_prg(1500, 60, 3)
    reads = tandem_reads(unit, site_at, alt, 100, 4)
    seeds = master_seeds(9, [len(reads)])
    want = oracle_map(prg, 8, reads, seeds)
    got, (n_over, n_cover_over, n_cover_huge), rc = hostemu_map(prg, 8, reads, seeds, big_states=1 << 16, big_arena=1 << 17)
    assert rc == 0
    assert n_cover_huge >= 4
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_copies", [300, 6000], ids=["large-capacity-tier", "last-tier"])
def test_gpu_many_instances_match_oracle(n_copies):
    """300 copies: the large-capacity search pass and the largest fixed scratch. 6 000 copies (>= 5 000 mapping
    instances per read): the heap-backed last tier, search and selection."""
    prg, unit, site_at, alt = tandem_prg(n_copies, 60, 5)
    reads = tandem_reads(unit, site_at, alt, 100, 6)
    seeds = master_seeds(11, [len(reads)])
    want = oracle_map(prg, 8, reads, seeds, threads=8)
    qm = Quasimapper(Index(prg, 8))
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    counts = qm.queue_counts()
    assert canonical_cov(qm.coverage()) == want
    if n_copies == 300:
        assert counts["big_mapped"] + counts["cover_overflow"] > 0 and counts["huge_search"] + counts["huge_cover"] == 0
    else:
        assert counts["huge_search"] + counts["huge_cover"] >= 4


@pytest.mark.gpu
def test_gpu_heap_exhaustion_is_reported():
    """With a heap far too small for 6 000 instances the engine names the read and returns GMX_ECAP — never a silent
    partial result: nothing of the read is recorded."""
    prg, unit, site_at, alt = tandem_prg(6000, 60, 5)
    reads = tandem_reads(unit, site_at, alt, 100, 6)[:2]
    seeds = master_seeds(11, [len(reads)])
    qm = Quasimapper(Index(prg, 8), huge_heap_bytes=64 * 1024)
    flat, offs = flatten_reads(reads)
    with pytest.raises(GmxError) as err:
        qm.map_reads(flat, offs, seeds)
    assert err.value.code == -4 and "huge_heap_bytes" in str(err.value)
    cov = qm.coverage()
    assert int(cov.raw_allele_sum.sum()) == 0 and int(cov.raw_per_base.sum()) == 0
