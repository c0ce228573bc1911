"""Host logic of the product (index builder) against the oracle: SA/BWT/rank, graph tables, jump programs, seeds."""
import numpy as np
import pytest

from oracle import Oracle
from gramtools_amd import Index, GmxError
from gramtools_amd.synth import nested_prg, bracket_to_ints, random_ref, snp_prg
from golden_runner import all_cases, prg_ints


def _multiset(states):
    return sorted((lo, hi, tuple(tvd), tuple(tvg)) for lo, hi, tvd, tvg in states)


def _check_index(prg, k):
    o = Oracle(prg, k)
    ix = Index(prg, k, threads=2)
    n = o.text_size()
    assert ix.sa().tolist() == o.sa().tolist()
    bwt = o.bwt()
    assert ix.bwt().tolist() == bwt.tolist()
    for i in sorted(set(list(range(0, n + 1, max(1, n // 97))) + [n])):
        for c in (1, 2, 3, 4):
            assert ix.rank(i, c) == o.rank(i, c), (i, c)
    # graph: node ids, offsets and marker targets per PRG position (coverage_graph.cpp:131-144,268-311)
    ra = o.random_access()
    pi = ix.pos_info()
    prg_arr = np.asarray(prg)
    assert pi[:, 0].tolist() == ra[:, 4].tolist() and pi[:, 1].tolist() == ra[:, 5].tolist()
    base = prg_arr <= 4
    assert pi[base, 2].tolist() == ra[base, 1].tolist()
    assert pi[:, 3].tolist() == ra[:, 2].tolist()
    has_t = ra[:, 2] != 0
    assert pi[has_t, 4].tolist() == ra[has_t, 3].tolist()
    assert ix.target_map() == o.target_map()
    pm = o.par_map()
    got = {5 + 2 * s: (int(ix.parent_site[s]), int(ix.parent_allele[s])) for s in range(ix.n_sites) if ix.parent_site[s]}
    assert got == pm
    assert ix.is_nested == o.is_nested() and ix.n_sites == o.num_sites()
    assert [len(a) for a in o.allele_sum()] == ix.n_alleles.tolist()
    # per-base layout covers exactly the oracle's coverage-owning nodes
    lay = {int(r[2]): int(r[4]) for r in ix.per_base_layout()}
    assert lay == {nd["first_pos"]: len(nd["cov"]) for nd in o.per_base_nodes()}
    # jump programs == search_state_vBWT_jumps on every single-position base interval and on each base's full interval
    sa = o.sa()
    for i in range(n):
        if sa[i] < len(prg) and prg[sa[i]] <= 4:
            assert _multiset(ix.jump_states(i, i)) == _multiset(o.vbwt_jumps((i, i, [], []))), i
    for c in (1, 2, 3, 4):
        idx = [i for i in range(n) if sa[i] < len(prg) and prg[sa[i]] == c]
        if idx:
            lo, hi = min(idx), max(idx)
            assert _multiset(ix.jump_states(lo, hi)) == _multiset(o.vbwt_jumps((lo, hi, [], [])))
    # seed table == k-mer index (all 4^k k-mers)
    if k:
        for km in Oracle.all_kmers(k):
            a, b = ix.seed_states(km), o.kmer_states(km)
            assert (a is None) == (b is None), km
            if a is not None:
                assert _multiset(a) == _multiset(b), km
    # the longer seed table == the reference's k-mer index for k2 (the same construction, continued)
    k2 = ix.info.kmer_size2
    if k2 and k2 <= 6:
        o2 = Oracle(list(prg), k2)
        for km in Oracle.all_kmers(k2):
            a, b = ix.seed_states(km), o2.kmer_states(km)
            assert (a is None) == (b is None), km
            if a is not None:
                assert _multiset(a) == _multiset(b), km


GOLDEN_PRGS = sorted({tuple(prg_ints(c["prg"])) for _, c in all_cases() if not c.get("expect_build_error")})


@pytest.mark.parametrize("prg", GOLDEN_PRGS, ids=[f"golden{i}" for i in range(len(GOLDEN_PRGS))])
def test_index_matches_oracle_on_golden_prgs(prg):
    try:
        Oracle(list(prg), 0).allele_sum()
    except Exception:
        pytest.skip("graph-only vector")
    if max(prg) > 4 and sorted({m for m in prg if m > 4 and m % 2}) != list(range(5, 5 + 2 * len({m for m in prg if m > 4 and m % 2}), 2)):
        pytest.skip("sparse site numbering: coverage layout undefined in the reference too")
    _check_index(list(prg), 2)


@pytest.mark.parametrize("seed", range(25))
def test_index_matches_oracle_on_random_nested_prgs(seed):
    rng = np.random.default_rng(seed)
    s = nested_prg(seed, n_top=int(rng.integers(1, 5)), max_depth=int(rng.integers(1, 4)), seq_max=int(rng.integers(1, 6)))
    if seed % 3 == 0:
        s = s.replace("t", "a").replace("g", "c")
    _check_index(bracket_to_ints(s).tolist(), int(rng.integers(1, 4)))


def test_inconsistent_prgs_are_rejected():
    for _, c in all_cases():
        if c.get("expect_build_error"):
            with pytest.raises(GmxError):
                Index(prg_ints(c["prg"]), 0)


def test_suffix_array_on_larger_snp_prg():
    ref = random_ref(20000, 1)
    prg, *_ = snp_prg(ref, 300, 2, multi_allelic_frac=0.2)
    o = Oracle(prg, 0)
    ix = Index(prg, 5, threads=2)
    assert (ix.sa() == o.sa()).all()
    assert ix.info.n_kmers_present > 900


def test_suffix_array_builder_with_the_index_types_top_bit_in_use():
    """configs[4] (whole human, 3.46 G symbols) needs suffix-array indices above 2^31: the builder is templated on an
    UNSIGNED index type (uint32_t in the index, as the reference's SA_Index). Instantiated with uint16_t it is run on
    texts of 33 000 .. 65 000 symbols — every index comparison, free-slot mark and loop bound then works above the
    signed range — against a plain sort of the suffixes."""
    import ctypes as C
    from gramtools_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    for n, alphabet in ((33000, 4), (50000, 2), (65533, 6)):
        text = rng.integers(1, alphabet + 1, size=n, dtype=np.uint16)
        # a few long repeats so that the recursion (names < LMS count) is taken more than once
        text[1000:9000] = text[20000:28000]
        text[-1] = 0
        out = np.zeros(n, dtype=np.uint16)
        assert lib.gmx_debug_suffix_array_u16(text.ctypes.data_as(C.POINTER(C.c_uint16)), n,
                                              out.ctypes.data_as(C.POINTER(C.c_uint16))) == 0
        raw = text.astype(np.uint8).tobytes()
        want = sorted(range(n), key=lambda i: raw[i:])
        assert out.astype(np.int64).tolist() == want


def test_parallel_suffix_sort_equals_sais(monkeypatch):
    """Round 3's parallel suffix sort (bucket by the class of the first eight symbols, then plain suffix comparisons; gives
    up on long repeats) against SA-IS on PRGs with sites, 10-copy repeats and a pathological period-4 text."""
    import numpy as np
    from gramtools_amd import Index
    from gramtools_amd.synth import random_ref, snp_prg
    rng = np.random.default_rng(4)
    ref = random_ref(60000, 21)
    seg = ref[1000:3000].copy()
    for dst in rng.integers(0, ref.size - 2000, size=10):
        ref[int(dst):int(dst) + 2000] = seg                      # repeats: long common prefixes
    prg, *_ = snp_prg(ref, 2000, 22, multi_allelic_frac=0.2)
    period = np.tile(np.array([1, 2, 3, 4], dtype=np.uint32), 30000)   # the comparison budget runs out: SA-IS takes over
    for text in (prg, period):
        monkeypatch.setenv("GMX_SAIS", "1")
        want = Index(text, 4, threads=4).sa()
        monkeypatch.delenv("GMX_SAIS")
        monkeypatch.setenv("GMX_PSORT_MIN", "1")
        got = Index(text, 4, threads=4).sa()
        assert np.array_equal(got, want)
