"""Index cache (gmx_index_save / gmx_index_load): a loaded index is the built one, table for table; a
cache of another PRG, another k, a truncated or corrupted file is refused; `gram build` writes it and `gram genotype`
uses it (gpu)."""
import os
import subprocess

import numpy as np
import pytest

from gramtools_amd import Index, Quasimapper, master_seeds
from gramtools_amd._lib import GmxError
from gramtools_amd.build import build_gram
from gramtools_amd.synth import bracket_to_ints, flat_offsets, nested_prg, random_ref, simulate_snp_reads, snp_prg

from common import canonical_cov


def _write_prg(path, prg):
    np.asarray(prg, dtype="<u4").tofile(path)


def _tables(ix):
    return (ix.sa().tolist(), ix.bwt().tolist(), ix.pos_info().tolist(), ix.target_map(), ix.per_base_layout().tolist(),
            ix.n_alleles.tolist(), ix.allele_sum_off.tolist(), ix.grouped_off.tolist(), ix.parent_site.tolist())


@pytest.mark.parametrize("seed", range(3))
def test_loaded_index_equals_built_index(tmp_path, seed):
    prg = bracket_to_ints(nested_prg(seed, n_top=5, max_depth=3)) if seed else snp_prg(random_ref(3000, 1), 40, 2)[0]
    prg_path, cache = str(tmp_path / "prg"), str(tmp_path / "cache.bin")
    _write_prg(prg_path, prg)
    built = Index(prg_path, 4)
    built.save(cache)
    loaded = Index(prg_path, 4, cache=cache)
    assert loaded.from_cache
    assert _tables(loaded) == _tables(built)
    kmers = [np.array(k, dtype=np.uint8) for k in ([1, 2, 3, 4], [4, 4, 1, 1], [2, 2, 2, 2])]
    for k in kmers:
        assert loaded.seed_states(k) == built.seed_states(k)
    assert loaded.info.index_bytes == built.info.index_bytes


def test_stale_or_damaged_cache_is_refused(tmp_path):
    prg = snp_prg(random_ref(2000, 1), 20, 2)[0]
    prg_path, cache = str(tmp_path / "prg"), str(tmp_path / "cache.bin")
    _write_prg(prg_path, prg)
    Index(prg_path, 5).save(cache)
    lib = Index(prg_path, 5).lib
    import ctypes as C
    h = C.c_void_p()
    assert lib.gmx_index_load(cache.encode(), prg_path.encode(), 6, C.byref(h)) != 0      # another k
    other = prg.copy()
    other[7] = 1 + other[7] % 4
    _write_prg(str(tmp_path / "prg2"), other)
    assert lib.gmx_index_load(cache.encode(), str(tmp_path / "prg2").encode(), 5, C.byref(h)) != 0  # another PRG
    data = open(cache, "rb").read()
    open(cache, "wb").write(data[: len(data) // 2])
    assert lib.gmx_index_load(cache.encode(), prg_path.encode(), 5, C.byref(h)) != 0      # truncated
    flipped = bytearray(data)                      # same length, same table sizes, one bit of one table flipped
    flipped[len(flipped) // 3] ^= 0x10
    open(cache, "wb").write(bytes(flipped))
    assert lib.gmx_index_load(cache.encode(), prg_path.encode(), 5, C.byref(h)) != 0      # checksum
    open(cache, "wb").write(b"not a cache")
    assert lib.gmx_index_load(cache.encode(), prg_path.encode(), 5, C.byref(h)) != 0
    ix = Index(prg_path, 5, cache=cache)  # falls back to building
    assert not ix.from_cache and ix.n_sites == 20


def test_gram_build_writes_the_cache(tmp_path):
    prg = snp_prg(random_ref(2000, 3), 25, 4)[0]
    gram_dir = tmp_path / "gram"
    gram_dir.mkdir()
    _write_prg(str(gram_dir / "prg"), prg)
    out = subprocess.run([build_gram(), "build", "--gram_dir", str(gram_dir), "--kmer_size", "6"], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "Wrote index cache" in out.stdout
    assert Index(str(gram_dir / "prg"), 6, cache=str(gram_dir / "gmx_index.k6.bin")).from_cache


@pytest.mark.gpu
def test_mapping_with_a_loaded_index_equals_the_oracle(tmp_path):
    """The index cache (`gram build` -> gmx_index.k<K>.bin -> gmx_index_load) against the ORACLE, not against the index it
    was saved from: coverage mapped with the loaded index == coverage of the CPU restatement of the reference on the same
    PRG, reads and seeds — on a nested PRG with indels and multi-allelic sites as well as on a SNP PRG."""
    from common import oracle_map
    from gramtools_amd.synth import nested_prg, bracket_to_ints, simulate_graph_reads
    ref = random_ref(30000, 5)
    prg, pos, alts, n_alts = snp_prg(ref, 400, 6, multi_allelic_frac=0.1)
    reads = list(simulate_snp_reads(ref, pos, alts, n_alts, 3000, 150, 7))
    nprg = bracket_to_ints(nested_prg(31, n_top=14, max_depth=3, seq_max=8))
    nreads = simulate_graph_reads(nprg, 800, 30, 8)
    for k, p, rd in ((8, prg, reads), (4, nprg, nreads)):
        seeds = master_seeds(9, [len(rd)])
        prg_path, cache = str(tmp_path / f"prg{k}"), str(tmp_path / f"cache{k}.bin")
        _write_prg(prg_path, p)
        Index(prg_path, k).save(cache)
        loaded = Index(prg_path, k, cache=cache)
        assert loaded.from_cache
        qm = Quasimapper(loaded)
        flat = np.concatenate([np.asarray(r, dtype=np.uint8) for r in rd])
        offs = np.concatenate([[0], np.cumsum([len(r) for r in rd])]).astype(np.uint64)
        qm.map_reads(flat, offs, seeds)
        assert canonical_cov(qm.coverage()) == oracle_map(p, k, rd, seeds, threads=8)
