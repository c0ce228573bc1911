"""The per-lane device logic (gmx_core.h + gmx_cover.h), run sequentially on the host by the TEST-ONLY
build tests/hostemu, must reproduce the oracle's coverage bit for bit — including the overflow tiers."""
import numpy as np
import pytest

from common import oracle_map, hostemu_map
from gramtools_amd.synth import nested_prg, bracket_to_ints, simulate_graph_reads, random_ref, snp_prg, simulate_snp_reads


def _case(seed):
    rng = np.random.default_rng(seed)
    s = nested_prg(seed, n_top=int(rng.integers(1, 6)), max_depth=int(rng.integers(1, 4)), seq_max=int(rng.integers(1, 7)))
    if seed % 3 == 0:  # low-complexity PRGs: repeats, multi-mapping, seeded selection
        s = s.replace("t", "a").replace("g", "c")
    prg = bracket_to_ints(s)
    L, k = int(rng.integers(4, 25)), int(rng.integers(1, 5))
    reads = simulate_graph_reads(prg, 40, L, seed + 100)
    reads += [rng.integers(1, 5, size=L).astype(np.uint8) for _ in range(5)]
    reads.append(np.array([1, 2, 0, 3] * 3, dtype=np.uint8))  # non-ACGT symbol -> skipped
    reads.append(np.array([4, 4, 1, 2, 5, 1, 1, 2, 3, 4, 1, 78], dtype=np.uint8))
    reads.append(np.zeros(0, dtype=np.uint8))                   # empty read -> skipped
    reads = [r for r in reads if len(r) >= k or len(r) == 0]
    seeds = rng.integers(0, 2 ** 32, size=len(reads), dtype=np.uint64).astype(np.uint32)
    return prg, k, reads, seeds


@pytest.mark.parametrize("seed", range(90))
def test_device_logic_matches_oracle(seed):
    prg, k, reads, seeds = _case(seed)
    mode = seed % 2
    want = oracle_map(prg, k, reads, seeds, rng_mode=mode)
    caps = dict(fast_states=1, fast_arena=2) if seed % 5 == 0 else {}
    got, _, rc = hostemu_map(prg, k, reads, seeds, rng_mode=mode, **caps)
    assert rc == 0
    assert got == want


def test_device_logic_on_snp_prg_150bp():
    ref = random_ref(6000, 11)
    prg, pos, alts, n_alts = snp_prg(ref, 80, 12, multi_allelic_frac=0.2)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, 300, 150, 13)
    seeds = np.arange(300, dtype=np.uint32) * 7919
    want = oracle_map(prg, 7, list(reads), seeds)
    got, _, rc = hostemu_map(prg, 7, list(reads), seeds)
    assert rc == 0 and got == want
    assert want["stats"]["exact_mapped"] == 300


@pytest.mark.parametrize("seed", range(12))
def test_inline_sites_dense_snps_short_reads(seed, monkeypatch):
    """Sites whose alleles are single bases are resolved inside the text record, without their marker record (gmx_types.h:
    INLINE sites). Dense bi- to quad-allelic SNPs (a site every ~8 bases, many per 64-symbol record, some straddling
    record ends), reads of 12-70 bases that start, end and seed inside sites: the device logic with inline sites, the same
    with them switched off (GMX_NO_INLINE_SITES) and the oracle agree."""
    from gramtools_amd import Index
    rng = np.random.default_rng(seed)
    ref = random_ref(int(rng.integers(300, 1500)), seed + 40)
    prg, pos, alts, n_alts = snp_prg(ref, ref.size // int(rng.integers(6, 12)), seed + 41, multi_allelic_frac=0.4)
    L, k = int(rng.integers(12, 70)), int(rng.integers(2, 8))
    reads = list(simulate_snp_reads(ref, pos, alts, n_alts, 150, L, seed + 42))
    # reads that start / end exactly at allele bases and next to markers
    for p in pos[:20]:
        for s0 in (int(p) - L + 1, int(p), int(p) - 1, int(p) + 1 - L // 2):
            if 0 <= s0 and s0 + L <= ref.size:
                reads.append(ref[s0:s0 + L].copy())
    seeds = rng.integers(0, 2 ** 32, size=len(reads), dtype=np.uint64).astype(np.uint32)
    want = oracle_map(prg, k, reads, seeds)
    ix = Index(prg, k)
    assert ix.info.n_inline_sites > 0.5 * ix.n_sites
    got, _, rc = hostemu_map(prg, k, reads, seeds)
    assert rc == 0 and got == want
    monkeypatch.setenv("GMX_NO_INLINE_SITES", "1")
    assert Index(prg, k).info.n_inline_sites == 0
    got2, _, rc = hostemu_map(prg, k, reads, seeds)
    assert rc == 0 and got2 == want


@pytest.mark.parametrize("seed", range(0, 90, 3))
def test_wide_single_instance_routine_matches_oracle(seed):
    """gmx_cover_single_nested_wide (the routine of gmx_cover_one_kernel: one final state of width one, loci in scratch,
    no keys / classes / draw) in front of the general routine: same coverage as the oracle on nested and repetitive PRGs."""
    prg, k, reads, seeds = _case(seed)
    want = oracle_map(prg, k, reads, seeds, rng_mode=seed % 2)
    st = {}
    got, _, rc = hostemu_map(prg, k, reads, seeds, rng_mode=seed % 2, wide=True, stats=st)
    assert rc == 0 and got == want
    assert st["n_wide"] > 0 or seed % 3 == 0   # (the low-complexity PRGs may have multi-mapping reads only)
