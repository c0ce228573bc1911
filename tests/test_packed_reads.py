"""Reads handed over as bit planes (include/gmx.h: gmx_pack_reads, gmx_map_reads_packed_host): the host packer against a
plain numpy statement of the layout (CPU), and the packed feed against the byte feed and the oracle (GPU).
Reference semantics kept: encode_dna_bases (common/utils.cpp:73-92) — any non-ACGT symbol drops the whole read, which still
counts (skipped, both orientations) and keeps its seed (quasimap.cpp:109-113)."""
import numpy as np
import pytest

from common import oracle_map, canonical_cov, flatten_reads
from gramtools_amd import Index, Quasimapper, QuasimapperGroup, master_seeds, pack_reads
from gramtools_amd.synth import nested_prg, bracket_to_ints, simulate_graph_reads, random_ref, snp_prg, simulate_snp_reads


def plane_words(read):
    """[(lo, hi)] per 32 bases of one encoded read, as the header defines them."""
    out = []
    for i in range(0, len(read), 32):
        lo = hi = 0
        for j, x in enumerate(read[i:i + 32]):
            c = (int(x) - 1) & 0xFF
            lo |= (c & 1) << j
            hi |= ((c >> 1) & 1) << j
        out.append(lo | (hi << 32))
    return out


def ragged_reads(rng, n, lens=(0, 1, 5, 31, 32, 33, 64, 150, 151, 300)):
    reads = []
    for i in range(n):
        L = int(lens[i % len(lens)])
        r = rng.integers(1, 5, size=L).astype(np.uint8)
        if L and i % 7 == 3:
            r[rng.integers(0, L)] = [0, 5, 78][i % 3]  # an unencodable base
        reads.append(r)
    return reads


def test_pack_reads_offsets_form_matches_the_layout_of_the_header():
    rng = np.random.default_rng(5)
    reads = ragged_reads(rng, 203)
    flat, offs = flatten_reads(reads)
    offs = offs + 77  # only differences and the >> 5 layout matter: a batch may be a window of a larger buffer
    flat = np.concatenate([np.zeros(77, np.uint8), flat])
    for threads in (1, 3):
        pk = pack_reads(flat, offs, threads=threads)
        for r, rd in enumerate(reads):
            at = (int(offs[r]) >> 5) - (int(offs[0]) >> 5) + r
            clean = bool(len(rd) == 0 or (rd.min() >= 1 and rd.max() <= 4))
            assert bool(pk.skip[r]) == (not clean)
            if clean:
                assert [int(x) for x in pk.planes[at:at + (len(rd) + 31) // 32]] == plane_words(rd), r


def test_a_sub_range_of_a_packed_batch_is_a_packed_batch():
    rng = np.random.default_rng(6)
    reads = [rng.integers(1, 5, size=int(L)).astype(np.uint8) for L in rng.integers(20, 200, size=400)]
    flat, offs = flatten_reads(reads)
    whole = pack_reads(flat, offs)
    P = lambda r: (int(offs[r]) >> 5) - (int(offs[0]) >> 5) + r
    for a, b in ((0, 400), (37, 211), (399, 400)):
        part = pack_reads(flat, offs[a:b + 1])
        n_pairs = P(b) - P(a)
        assert np.array_equal(part.planes[:n_pairs], whole.planes[P(a):P(b)])


def test_pack_reads_uniform_form_is_back_to_back_and_checks_the_lengths():
    rng = np.random.default_rng(7)
    reads = rng.integers(1, 5, size=(300, 150)).astype(np.uint8)
    offs = (np.arange(301) * 150).astype(np.uint64)
    pk = pack_reads(reads.reshape(-1), offs, uniform_len=150, threads=2)
    assert pk.offsets is None and pk.uniform_len == 150
    for r in (0, 1, 150, 299):
        assert [int(x) for x in pk.planes[5 * r:5 * r + 5]] == plane_words(reads[r])
    from gramtools_amd import GmxError
    with pytest.raises(GmxError):
        pack_reads(reads.reshape(-1)[:-1], np.append(offs[:-1], offs[-1] - 1), uniform_len=150)


# ---------------------------------------------------------------------------------------------------------------------
def _snp_case(n_reads, seed):
    ref = random_ref(30000, seed)
    prg, pos, alts, n_alts = snp_prg(ref, 400, seed + 1)
    reads = simulate_snp_reads(ref, pos, alts, n_alts, n_reads, 150, seed + 2)
    return prg, reads


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_packed_feed_equals_byte_feed_equals_oracle_uniform(pinned):
    prg, reads = _snp_case(6000, 11)
    reads = reads.copy()
    reads[5, 17] = 0      # unencodable reads: skipped, keep their seed
    reads[4999, 149] = 9
    k = 7
    seeds = master_seeds(42, [len(reads)])
    offs = (np.arange(len(reads) + 1) * 150).astype(np.uint64)
    ix = Index(prg, k)
    a = Quasimapper(ix)
    a.map_reads(reads.reshape(-1), offs, seeds)
    want = canonical_cov(a.coverage())
    assert want == oracle_map(prg, k, list(reads), seeds)
    pk = pack_reads(reads.reshape(-1), offs, uniform_len=150, pinned=pinned)
    assert pk.skip[5] == 1 and pk.skip[4999] == 1 and int(pk.skip.sum()) == 2
    b = Quasimapper(ix, max_batch_reads=1000)   # six chunks through the three upload slots
    b.map_reads_packed(pk, seeds)
    b.map_reads_packed(pk, seeds)               # a second call while the first may still be in flight
    b.sync()
    got = canonical_cov(b.coverage())
    twice = Quasimapper(ix)
    twice.map_reads(reads.reshape(-1), offs, seeds)
    twice.map_reads(reads.reshape(-1), offs, seeds)
    assert got == canonical_cov(twice.coverage())
    c = Quasimapper(ix)
    c.map_reads_packed(pk, seeds)
    assert canonical_cov(c.coverage()) == want
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_packed_feed_ragged_reads_nested_prg(pinned):
    rng = np.random.default_rng(3)
    prg = bracket_to_ints(nested_prg(17, n_top=12, max_depth=2, seq_max=8))
    k = 4
    good = simulate_graph_reads(prg, 1500, 40, 21)
    reads = []
    for i, r in enumerate(good):
        r = np.asarray(r, dtype=np.uint8)
        if i % 5 == 0:
            r = r[:int(rng.integers(k, len(r) + 1))]          # ragged
        if i % 11 == 0 and len(r):
            r = r.copy()
            r[int(rng.integers(0, len(r)))] = 0               # an N
        reads.append(r)
    reads.append(np.zeros(0, dtype=np.uint8))                 # an empty read
    seeds = master_seeds(7, [len(reads)])
    want = oracle_map(prg, k, reads, seeds)
    flat, offs = flatten_reads(reads)
    ix = Index(prg, k)
    a = Quasimapper(ix)
    a.map_reads(flat, offs, seeds)
    assert canonical_cov(a.coverage()) == want
    pk = pack_reads(flat, offs, pinned=pinned)
    b = Quasimapper(ix, max_batch_reads=400)
    b.map_reads_packed(pk, seeds)
    assert canonical_cov(b.coverage()) == want
    pk.close()
    # reads shorter than k (undefined in the reference, quasimap.cpp:206-210: counted as skipped here) — both feeds agree
    short = reads + [np.asarray(r[:k - 1], dtype=np.uint8) for r in reads[:50] if len(r) >= k]
    seeds2 = master_seeds(8, [len(short)])
    flat2, offs2 = flatten_reads(short)
    c = Quasimapper(ix)
    c.map_reads(flat2, offs2, seeds2)
    pk2 = pack_reads(flat2, offs2, pinned=pinned)
    d = Quasimapper(ix)
    d.map_reads_packed(pk2, seeds2)
    assert canonical_cov(d.coverage()) == canonical_cov(c.coverage())
    assert c.coverage().stats.skipped_reads_count >= 100
    pk2.close()


@pytest.mark.gpu
def test_packed_feed_through_a_group_of_engines():
    prg, reads = _snp_case(5000, 31)
    k = 7
    seeds = master_seeds(9, [len(reads)])
    offs = (np.arange(len(reads) + 1) * 150).astype(np.uint64)
    want = oracle_map(prg, k, list(reads), seeds)
    ix = Index(prg, k)
    for uniform in (150, 0):
        pk = pack_reads(reads.reshape(-1), offs, uniform_len=uniform, pinned=True)
        g = QuasimapperGroup(ix, [0, 0, 0])
        g.map_reads_packed(pk, seeds)
        g.allreduce()
        assert canonical_cov(g.coverage(0)) == want
        assert canonical_cov(g.coverage(2)) == want
        g.close()
        pk.close()


@pytest.mark.gpu
def test_seeds_read_in_place_give_the_same_draws():
    """gmx_engine_seeds_in_place: no seeds are uploaded, the kernels read the ones they need from the caller's page-locked
    buffer. A nested PRG with repeats, so that many reads have several equally good classes and DRAW
    (coverage_common.cpp:166-177): the coverage must be the oracle's, and differ from a run with other seeds."""
    from gramtools_amd import PinnedArray
    prg = bracket_to_ints(nested_prg(23, n_top=10, max_depth=2, seq_max=6).replace("t", "a"))
    k = 4
    reads = [np.asarray(r, dtype=np.uint8) for r in simulate_graph_reads(prg, 3000, 18, 5)]
    seeds = master_seeds(11, [len(reads)])
    want = oracle_map(prg, k, reads, seeds)
    assert want != oracle_map(prg, k, reads, master_seeds(12, [len(reads)])), "the case must depend on its seeds"
    flat, offs = flatten_reads(reads)
    ix = Index(prg, k)
    pk = pack_reads(flat, offs, pinned=True)
    sd = PinnedArray(len(reads), np.uint32)
    sd.array[:] = seeds
    for batch in (None, 700):
        qm = Quasimapper(ix) if batch is None else Quasimapper(ix, max_batch_reads=batch)
        qm.seeds_in_place(True)
        qm.map_reads_packed(pk, sd.array)
        assert canonical_cov(qm.coverage()) == want
    # seeds in pageable memory with the switch on: uploaded as before
    qm = Quasimapper(ix)
    qm.seeds_in_place(True)
    qm.map_reads_packed(pk, seeds)
    assert canonical_cov(qm.coverage()) == want
    pk.close()
    sd.close()


def test_two_bit_stream_layout():
    """gmx_pack_reads_2bit: base j of the batch in bits 2j, 2j + 1 — the reads back to back, whatever their lengths."""
    from gramtools_amd import pack_reads_2bit
    rng = np.random.default_rng(9)
    reads = [rng.integers(1, 5, size=int(L)).astype(np.uint8) for L in rng.integers(0, 200, size=300)]
    reads[5] = reads[5].copy()
    if len(reads[5]) == 0:
        reads[5] = np.array([1, 2, 3], dtype=np.uint8)
    reads[5][0] = 0
    flat, offs = flatten_reads(reads)
    for threads in (1, 4):
        pk = pack_reads_2bit(flat, offs, threads=threads)
        bits = np.unpackbits(pk.planes.view(np.uint8), bitorder="little")
        codes = bits[0:2 * flat.size:2] | (bits[1:2 * flat.size:2] << 1)
        ok = np.ones(flat.size, dtype=bool)
        ok[int(offs[5]):int(offs[6])] = False  # a skipped read's bits are not defined
        assert (codes[ok] == (flat[ok] - 1)).all()
        assert pk.skip[5] == 1 and pk.skip[:5].sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_two_bit_stream_feed_equals_oracle(pinned):
    from gramtools_amd import pack_reads_2bit
    # uniform reads on a SNP PRG, several chunks per call
    prg, reads = _snp_case(6000, 41)
    k = 7
    seeds = master_seeds(13, [len(reads)])
    want = oracle_map(prg, k, list(reads), seeds, threads=8)
    flat, offs = flatten_reads(list(reads))
    ix = Index(prg, k)
    for batch in (None, 1100):
        pk = pack_reads_2bit(flat, offs, uniform_len=reads.shape[1], pinned=pinned)
        qm = Quasimapper(ix) if batch is None else Quasimapper(ix, max_batch_reads=batch)
        qm.map_reads_packed(pk, seeds)
        assert canonical_cov(qm.coverage()) == want
        pk.close()
    # ragged reads with Ns and an empty read on a nested PRG
    rng = np.random.default_rng(3)
    prg = bracket_to_ints(nested_prg(17, n_top=12, max_depth=2, seq_max=8))
    k = 4
    rr = []
    for i, r in enumerate(simulate_graph_reads(prg, 1500, 40, 21)):
        r = np.asarray(r, dtype=np.uint8)
        if i % 5 == 0:
            r = r[:int(rng.integers(k, len(r) + 1))]
        if i % 11 == 0 and len(r):
            r = r.copy()
            r[int(rng.integers(0, len(r)))] = 0
        rr.append(r)
    rr.append(np.zeros(0, dtype=np.uint8))
    seeds = master_seeds(7, [len(rr)])
    want = oracle_map(prg, k, rr, seeds)
    flat, offs = flatten_reads(rr)
    pk = pack_reads_2bit(flat, offs, pinned=pinned)
    qm = Quasimapper(Index(prg, k), max_batch_reads=400)
    qm.map_reads_packed(pk, seeds)
    assert canonical_cov(qm.coverage()) == want
    pk.close()
