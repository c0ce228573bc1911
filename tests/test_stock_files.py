"""readers of the SDSL-lite 2.1.1 vectors a stock `gramtools build` leaves in gram_dir — kmers, kmers_stats,
sa_intervals, paths (build/kmer_index/load.cpp:71-173; written by dump.cpp:27-137) and the four base masks
(prg/make_data_structures.cpp:78-156).

PARITY UNPINNED: the reference holds no such file (its tests build them in memory) and cannot be built here, so nothing
below compares with bytes the reference wrote. Checked instead: (1) the on-disk form as sdsl/int_vector.hpp of v2.1.1
specifies it, on byte strings assembled here by hand; (2) the layout rules of dump.cpp / load.cpp on a k-mer index small
enough to write down; (3) a round trip: the native index written in those formats, read back the way load.cpp reads them,
compared with the native index (and failing when a value is changed)."""
import ctypes as C
import struct

import numpy as np
import pytest

from gramtools_amd import Index, _lib
from gramtools_amd.synth import bracket_to_ints, nested_prg, random_ref, snp_prg
from oracle import Oracle


def _read(path, fixed_width):
    lib = _lib.load()
    w = C.c_uint32(0)
    n = lib.gmx_stock_read_int_vector(str(path).encode(), fixed_width, None, 0, C.byref(w))
    assert n >= 0, lib.gmx_last_error()
    out = np.zeros(max(n, 1), dtype=np.uint64)
    assert lib.gmx_stock_read_int_vector(str(path).encode(), fixed_width, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size, C.byref(w)) == n
    return out[:n].tolist(), w.value


def test_sdsl_int_vector_bytes_assembled_by_hand(tmp_path):
    # int_vector<>: 64-bit length in BITS, one width byte, values LSB-first in 64-bit words. Five values of width 13.
    vals = [1, 8191, 4660, 0, 4095]
    stream = sum(v << (13 * i) for i, v in enumerate(vals))
    (tmp_path / "v13").write_bytes(struct.pack("<QB", 65, 13) + struct.pack("<QQ", stream & (2 ** 64 - 1), stream >> 64))
    assert _read(tmp_path / "v13", 0) == (vals, 13)
    # int_vector<3> (kmers): no width byte; 22 values -> 66 bits -> two words
    bases = [1, 2, 3, 4] * 5 + [4, 1]
    stream = sum(v << (3 * i) for i, v in enumerate(bases))
    (tmp_path / "k3").write_bytes(struct.pack("<Q", 66) + struct.pack("<QQ", stream & (2 ** 64 - 1), stream >> 64))
    assert _read(tmp_path / "k3", 3) == (bases, 3)
    # bit_vector: 70 bits
    bits = [(i * 7 + 3) % 5 == 0 for i in range(70)]
    stream = sum(int(b) << i for i, b in enumerate(bits))
    (tmp_path / "b").write_bytes(struct.pack("<Q", 70) + struct.pack("<QQ", stream & (2 ** 64 - 1), stream >> 64))
    assert _read(tmp_path / "b", 1) == ([int(b) for b in bits], 1)
    # a truncated file and an impossible width are errors, not garbage
    (tmp_path / "short").write_bytes(struct.pack("<QB", 6400, 32) + b"\0" * 16)
    lib = _lib.load()
    assert lib.gmx_stock_read_int_vector(str(tmp_path / "short").encode(), 0, None, 0, None) < 0
    (tmp_path / "w0").write_bytes(struct.pack("<QB", 64, 0) + b"\0" * 8)
    assert lib.gmx_stock_read_int_vector(str(tmp_path / "w0").encode(), 0, None, 0, None) < 0


def test_writer_and_reader_agree_on_every_width(tmp_path):
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for width in (1, 3, 7, 13, 31, 32, 33, 63, 64):
        vals = rng.integers(0, 2 ** 63, size=257, dtype=np.uint64) & np.uint64((1 << width) - 1 if width < 64 else 2 ** 64 - 1)
        for fixed in (0, 1):
            p = tmp_path / f"w{width}_{fixed}"
            assert lib.gmx_stock_write_int_vector(str(p).encode(), vals.ctypes.data_as(C.POINTER(C.c_uint64)), vals.size, width, fixed) == 0
            got, w = _read(p, width if fixed else 0)
            assert w == width and got == vals.tolist()
            assert p.stat().st_size == 8 + (0 if fixed else 1) + 8 * ((vals.size * width + 63) // 64)


def test_kmer_index_files_written_down_by_hand(tmp_path):
    """PRG a5g6t6c (1 5 3 6 4 6 2), k = 2: the states of every 2-mer from the oracle (pinned by the reference's own vectors),
    laid out by hand as dump.cpp lays them out — k-mers in an arbitrary order, as the reference's hash map would — and read."""
    prg = [1, 5, 3, 6, 4, 6, 2]
    o = Oracle(prg, 2, all_kmers=True)
    entries = []
    for a in (4, 3, 2, 1):          # (not the table order)
        for b in (1, 3, 2, 4):
            st = o.kmer_states(np.array([a, b], dtype=np.uint8))
            if st:
                entries.append(([a, b], st))
    assert len(entries) >= 4
    kmers, stats, sa, paths = [], [], [], []
    for km, st in entries:
        kmers += km
        stats.append(len(st))
        for lo, hi, tvd, tvg in st:
            stats.append(len(tvd) + len(tvg))
            sa += [lo, hi]
            for m, al in tvd:
                paths += [m, al + 1]
            for m, _ in tvg:
                paths += [m, 0]
    lib = _lib.load()

    def put(name, vals, width, fixed):
        v = np.asarray(vals if len(vals) else [0], dtype=np.uint64)
        assert lib.gmx_stock_write_int_vector(str(tmp_path / name).encode(), v.ctypes.data_as(C.POINTER(C.c_uint64)), len(vals), width, fixed) == 0
    put("kmers", kmers, 3, 1)
    put("kmers_stats", stats, max(1, max(stats).bit_length()), 0)
    put("sa_intervals", sa, max(1, max(sa).bit_length()), 0)
    put("paths", paths, max(1, max(paths + [0]).bit_length()), 0)
    bwt = o.bwt().tolist()
    for base, name in enumerate("acgt", start=1):
        put(f"{name}_base_bwt_mask", [int(x == base) for x in bwt], 1, 1)
    ix = Index(prg, 2)
    rep = _lib.StockReport()
    assert lib.gmx_index_check_stock_files(ix.h, str(tmp_path).encode(), C.byref(rep)) == 0, lib.gmx_last_error()
    assert rep.kmers == len(entries) and rep.states == sum(len(st) for _, st in entries)
    assert rep.kmer_mismatches == 0 and rep.kmers_missing_in_files == 0 and rep.duplicate_kmers == 0
    assert rep.mask_bits == 4 * len(bwt) and rep.mask_mismatches == 0


@pytest.mark.parametrize("seed", range(6))
def test_round_trip_of_the_native_index(tmp_path, seed):
    if seed < 4:
        prg = bracket_to_ints(nested_prg(seed + 50, n_top=4, max_depth=3, seq_max=6))
        k = 3
    else:
        ref = random_ref(4000, seed)
        prg, *_ = snp_prg(ref, 120, seed + 1, multi_allelic_frac=0.3)
        k = 5
    ix = Index(prg, k)
    lib = _lib.load()
    assert lib.gmx_index_write_stock_files(ix.h, str(tmp_path).encode()) == 0, lib.gmx_last_error()
    rep = _lib.StockReport()
    assert lib.gmx_index_check_stock_files(ix.h, str(tmp_path).encode(), C.byref(rep)) == 0, lib.gmx_last_error()
    assert rep.kmers > 0 and rep.states >= rep.kmers and rep.kmer_mismatches == 0 and rep.kmers_missing_in_files == 0
    assert rep.mask_bits == 4 * (len(prg) + 1) and rep.mask_mismatches == 0
    # the files hold what load.cpp expects: kmers is k symbols per k-mer, the masks partition the non-marker BWT positions
    kmers, w = _read(tmp_path / "kmers", 3)
    assert w == 3 and len(kmers) == k * rep.kmers and set(kmers) <= {1, 2, 3, 4}
    masks = [np.array(_read(tmp_path / f"{c}_base_bwt_mask", 1)[0]) for c in "acgt"]
    assert (sum(masks) <= 1).all() and int(sum(masks).sum()) == sum(1 for x in prg if x <= 4)
    # a changed SA interval is noticed
    sa, w = _read(tmp_path / "sa_intervals", 0)
    sa[0] ^= 1
    v = np.asarray(sa, dtype=np.uint64)
    assert lib.gmx_stock_write_int_vector(str(tmp_path / "sa_intervals").encode(), v.ctypes.data_as(C.POINTER(C.c_uint64)), v.size, w, 0) == 0
    assert lib.gmx_index_check_stock_files(ix.h, str(tmp_path).encode(), C.byref(rep)) == 0
    assert rep.kmer_mismatches == 1


def test_gram_build_write_and_check_stock(tmp_path):
    """`gram build --write_stock` then `gram build --check_stock` (round 6: the readers are wired into the executable): the
    round trip agrees; a cov_graph whose Boost archive head carries the right / a wrong site count is told apart (hand-assembled
    head: signature string, library version 17, class information, bubble_map's element count — unpinned like the rest); a
    damaged mask makes the command fail with exit code 1."""
    import subprocess
    from gramtools_amd.build import build_gram
    gram = build_gram()
    ref = random_ref(3000, 9)
    prg, *_ = snp_prg(ref, 80, 10, multi_allelic_frac=0.3)
    n_sites = len({x for x in prg if x > 4 and x % 2 == 1})
    np.asarray(prg, dtype="<u4").tofile(tmp_path / "prg")

    def run(*flags):
        return subprocess.run([gram, "build", "--gram_dir", str(tmp_path), "--kmer_size", "5", "--max_threads", "2", *flags],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    r = run("--write_stock")
    assert r.returncode == 0 and "Wrote kmers" in r.stdout, r.stdout
    r = run("--check_stock")
    assert r.returncode == 0 and "Stock files agree" in r.stdout and "cov_graph: absent" in r.stdout, r.stdout

    def archive_head(count):
        sig = b"serialization::archive"
        return struct.pack("<Q", len(sig)) + sig + struct.pack("<H", 17) + b"\x00\x00\x00\x00\x00" + struct.pack("<Q", count) + b"\x00" * 40
    (tmp_path / "cov_graph").write_bytes(archive_head(n_sites))
    (tmp_path / "fm_index").write_bytes(b"\x00" * 100)
    r = run("--check_stock")
    assert r.returncode == 0 and "archive signature and site count agree (Boost archive version 17)" in r.stdout and "fm_index: 100 bytes" in r.stdout, r.stdout
    (tmp_path / "cov_graph").write_bytes(archive_head(n_sites + 3))
    r = run("--check_stock")
    assert r.returncode == 1 and "site count differs" in r.stdout and "DIFFER" in r.stdout, r.stdout
    (tmp_path / "cov_graph").unlink()
    bits, w = _read(tmp_path / "g_base_bwt_mask", 1)
    bits[7] ^= 1
    v = np.asarray(bits, dtype=np.uint64)
    assert _lib.load().gmx_stock_write_int_vector(str(tmp_path / "g_base_bwt_mask").encode(), v.ctypes.data_as(C.POINTER(C.c_uint64)), v.size, 1, 1) == 0
    r = run("--check_stock")
    assert r.returncode == 1 and "1 differ" in r.stdout, r.stdout
