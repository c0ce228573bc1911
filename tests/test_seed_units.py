"""Multi-state k-mer index entries addressed in units of 2^seed_shift words (gmx_types.h GmxSeed; the builder picks the
shift from 2^30 words on — whole-genome PRGs, BASELINE configs[4] — and GMX_SEED_SHIFT forces it here). The entries
themselves (build/kmer_index/build.cpp:101-131) and everything mapped from them must not depend on the shift."""
import numpy as np
import pytest

from common import oracle_map, canonical_cov, flatten_reads, hostemu_map
from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd.synth import nested_prg, bracket_to_ints, simulate_graph_reads, random_ref, mixed_variant_prg, simulate_haplotype_reads


def _nested_case(seed):
    prg = bracket_to_ints(nested_prg(seed + 40, n_top=10, max_depth=3).replace("t", "a"))  # repeats: many multi-state entries
    reads = simulate_graph_reads(prg, 300, 16, seed)
    return prg, reads, master_seeds(seed, [len(reads)])


@pytest.mark.parametrize("shift", [1, 4])
def test_entries_are_the_same_states_under_any_unit(monkeypatch, shift):
    prg, _, _ = _nested_case(1)
    plain = Index(prg, 4, threads=1)
    monkeypatch.setenv("GMX_SEED_SHIFT", str(shift))
    units = Index(prg, 4, threads=1)
    n_multi = 0
    import itertools
    for kmer in itertools.product((1, 2, 3, 4), repeat=4):
        a, b = plain.seed_states(kmer), units.seed_states(kmer)
        assert a == b
        n_multi += a is not None and len(a) > 1
    assert n_multi > 20
    assert units.info.index_bytes > plain.info.index_bytes  # the padding


@pytest.mark.parametrize("slice_entries", [1, 3, 7, 1 << 16])
@pytest.mark.parametrize("shift", [2, 5])
def test_padding_in_place_does_not_depend_on_how_the_entries_are_sliced(monkeypatch, tmp_path, shift, slice_entries):
    """The entries are padded to their units inside the one buffer that holds them (no second copy: what lets the 85 M-site
    whole-genome PRG build inside the container's memory): slices from the end, overlapping ones entry by entry backwards,
    the others side by side. Whatever the slicing and the thread count, the index file has the same bytes."""
    ref = random_ref(3000, 9)
    prg, _ = mixed_variant_prg(ref, 130, 10, max_alleles=4)
    np.asarray(prg, dtype="<u4").tofile(str(tmp_path / "prg"))
    monkeypatch.setenv("GMX_SEED_SHIFT", str(shift))
    monkeypatch.setenv("GMX_SEED_SLICE", str(1 << 16))
    want = Index(prg, 6, threads=1)
    want.save(str(tmp_path / "want.gmx"))
    monkeypatch.setenv("GMX_SEED_SLICE", str(slice_entries))
    got = Index(prg, 6, threads=4)
    got.save(str(tmp_path / "got.gmx"))
    assert open(tmp_path / "got.gmx", "rb").read() == open(tmp_path / "want.gmx", "rb").read()
    import itertools
    monkeypatch.delenv("GMX_SEED_SHIFT")
    plain = Index(prg, 6, threads=2)
    n_multi = 0
    for kmer in itertools.product((1, 2, 3, 4), repeat=6):
        a = plain.seed_states(kmer)
        assert got.seed_states(kmer) == a
        n_multi += a is not None and len(a) > 1
    assert n_multi > 200


@pytest.mark.parametrize("seed", [1, 2])
def test_host_emulation_maps_the_same_with_units(monkeypatch, seed):
    prg, reads, seeds = _nested_case(seed)
    want = oracle_map(prg, 4, reads, seeds)
    monkeypatch.setenv("GMX_SEED_SHIFT", "3")
    got, _, rc = hostemu_map(prg, 4, reads, seeds)
    assert rc == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("cursor", ["0", "1"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_engine_maps_the_same_with_units(monkeypatch, seed, cursor):
    prg, reads, seeds = _nested_case(seed)
    want = oracle_map(prg, 4, reads, seeds)
    monkeypatch.setenv("GMX_SEED_SHIFT", "4")
    monkeypatch.setenv("GMX_SEED_CURSOR", cursor)  # entries taken state by state through the 64-bit cursor, or pushed whole
    qm = Quasimapper(Index(prg, 4))
    flat, offs = flatten_reads(reads)
    qm.map_reads(flat, offs, seeds)
    assert canonical_cov(qm.coverage()) == want


@pytest.mark.gpu
def test_engine_flat_prg_with_units_and_a_cached_index(monkeypatch, tmp_path):
    ref = random_ref(5000, 3)
    prg, sites = mixed_variant_prg(ref, 120, 4, max_alleles=4)
    reads = simulate_haplotype_reads(ref, sites, 2000, 60, 150, 5)
    seeds = master_seeds(42, [len(reads)])
    want = oracle_map(prg, 7, reads, seeds, threads=8)
    monkeypatch.setenv("GMX_SEED_SHIFT", "2")
    built = Index(prg, 7)
    prg_path, cache = str(tmp_path / "prg"), str(tmp_path / "ix.gmx")
    np.asarray(prg, dtype="<u4").tofile(prg_path)
    built.save(cache)
    monkeypatch.delenv("GMX_SEED_SHIFT")
    loaded = Index(prg_path, 7, cache=cache)
    assert loaded.from_cache
    flat, offs = flatten_reads(reads)
    for ix in (built, loaded):
        qm = Quasimapper(ix)
        qm.map_reads(flat, offs, seeds)
        assert canonical_cov(qm.coverage()) == want


@pytest.mark.gpu
@pytest.mark.parametrize("k,shift", [(5, "0"), (6, "3"), (4, "2")])
def test_screening_side_table_matches_oracle_and_the_header_walk(monkeypatch, k, shift):
    """Round 6: the seed cursor's screen reads one side word per state (six bases of left context + the state's offset;
    gmx_seed_side_kernel) instead of every state's header. A small k on a dense PRG (configs[4]'s variant mix: SNPs, anchored
    indels, pure deletions, 3-4 alleles, a site every 36 bases) gives every k-mer hundreds of states — text-form ones, interval
    states, states with paths (whose headers have other lengths: the offsets) —, reads error-free and as a sequencer delivers
    them (mismatches: the six bases pass, the fourteen reject; Ns; ragged). Against the oracle, and identical raw accumulators
    with GMX_NO_SEED_SIDE=1 (the header walk of rounds 4-5)."""
    from gramtools_amd.synth import chr20_recipe, realistic_reads, flat_offsets
    prg, reads2d = chr20_recipe(150_000, 4_100, 2_500, 70 + k)
    clean = (np.ascontiguousarray(reads2d).reshape(-1), flat_offsets(*reads2d.shape))
    cases = [clean, realistic_reads(reads2d, 9, sub_rate=0.01, n_read_frac=0.02, len_lo=60)]
    monkeypatch.setenv("GMX_SEED_SHIFT", shift)
    monkeypatch.setenv("GMX_SEED_CURSOR", "1")
    ix = Index(prg, k)
    for flat, offs in cases:
        seeds = master_seeds(42, [len(offs) - 1])
        want = oracle_map(prg, k, (flat, offs), seeds, threads=8)
        raws = []
        for no_side in (False, True):
            if no_side:
                monkeypatch.setenv("GMX_NO_SEED_SIDE", "1")
            else:
                monkeypatch.delenv("GMX_NO_SEED_SIDE", raising=False)
            qm = Quasimapper(ix)
            qm.map_reads(flat, offs, seeds)
            cov = qm.coverage()
            assert qm.queue_counts()["seed_cursor"] == 1
            assert canonical_cov(cov) == want, "header walk" if no_side else "side table"
            raws.append((cov.raw_allele_sum.copy(), cov.raw_per_base.copy(), cov.raw_grouped.copy(), cov.stats.as_dict()))
        assert all((a == b).all() for a, b in zip(raws[0][:3], raws[1][:3])) and raws[0][3] == raws[1][3]
