"""ctypes binding of oracle/quasimap_oracle.cpp — TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgmo_oracle.so")
_SRC = os.path.join(_HERE, "quasimap_oracle.cpp")

RNG_LEMIRE = 0    # libstdc++ >= 11 uniform_int_distribution (this toolchain)
RNG_DIVISION = 1  # libstdc++ <= 10


def build_oracle(force: bool = False) -> str:
    """Compile the oracle shared object if missing or stale. Returns its path."""
    stale = (not os.path.exists(_SO)) or (
        os.path.exists(_SRC) and os.path.getmtime(_SRC) > os.path.getmtime(_SO))
    if force or stale:
        subprocess.check_call(
            ["g++", "-O2", "-std=c++17", "-fopenmp", "-fPIC", "-shared",
             "-Wno-unknown-pragmas", "-o", _SO, _SRC])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build_oracle())
    vp, u64, u32, i64p = C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_int64)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    sig = {
        "gmo_create": (vp, [u32p, u64, u32, C.c_int, C.c_int]),
        "gmo_create_error": (C.c_char_p, []),
        "gmo_destroy": (None, [vp]),
        "gmo_last_error": (C.c_char_p, [vp]),
        "gmo_index_kmers": (C.c_int, [vp, u8p, u64]),
        "gmo_index_kmer_diffs": (C.c_int, [vp, u8p, u64p, u64]),
        "gmo_all_kmers": (None, [u32, u8p]),
        "gmo_prefix_diffs": (u64, [u8p, u64, u32, u8p, u64p]),
        "gmo_reverse_complement": (None, [u8p, u64, u8p]),
        "gmo_all_kmers_in_index": (C.c_int, [vp, u8p, u64]),
        "gmo_max_cov_haplogroup": (None, [vp, u64, i64p, i64p]),
        "gmo_extract_max_cov_allele": (C.c_long, [vp, u64, C.c_char_p, C.c_long, i64p]),
        "gmo_set_par_map": (C.c_int, [vp, i64p, u64]),
        "gmo_check_site_uniqueness": (C.c_int, [vp, i64p]),
        "gmo_assign_loci": (C.c_long, [vp, i64p, u64, i64p, i64p, C.c_long]),
        "gmo_unique_site_paths": (C.c_long, [vp, i64p, i64p, C.c_long]),
        "gmo_set_grouped": (None, [vp, u64, i64p, u64, u32]),
        "gmo_text_size": (u64, [vp]),
        "gmo_sa": (None, [vp, u32p]),
        "gmo_bwt": (None, [vp, u32p]),
        "gmo_rank": (u64, [vp, u64, u32]),
        "gmo_C_of": (u64, [vp, u32]),
        "gmo_marker_sa_interval": (C.c_int, [vp, u32, u32p, u32p]),
        "gmo_base_next_sa_interval": (C.c_int, [vp, u32, u32, u32, u32, u32p, u32p]),
        "gmo_left_markers_search": (C.c_long, [vp, u32, u32, i64p, C.c_long]),
        "gmo_vbwt_jumps": (C.c_long, [vp, i64p, i64p, C.c_long]),
        "gmo_search_base_backwards": (C.c_long, [vp, u32, i64p, i64p, C.c_long]),
        "gmo_process_read_char": (C.c_long, [vp, u32, i64p, i64p, C.c_long]),
        "gmo_kmer_states": (C.c_long, [vp, u8p, i64p, C.c_long]),
        "gmo_kmer_index_size": (u64, [vp]),
        "gmo_search_read_backwards": (C.c_long, [vp, u8p, u64, i64p, C.c_long]),
        "gmo_encapsulated": (C.c_long, [vp, i64p, i64p, C.c_long]),
        "gmo_locus_finder": (C.c_long, [vp, i64p, i64p, C.c_long]),
        "gmo_select_forced": (C.c_long, [vp, i64p, u32, i64p, C.c_long]),
        "gmo_rng_raw": (None, [u32, u32, u32p]),
        "gmo_rng_generate": (None, [u32, u32, u32, u32, C.c_int, u32p]),
        "gmo_rng_generate_std": (None, [u32, u32, u32, u32, u32p]),
        "gmo_master_seeds": (None, [u32, u64p, u64, u32p]),
        "gmo_quasimap_read": (C.c_int, [vp, u8p, u64, u32]),
        "gmo_map_reads": (C.c_int, [vp, u8p, u64p, u32p, u64, C.c_int]),
        "gmo_stats": (None, [vp, u64p]),
        "gmo_reset_coverage": (None, [vp]),
        "gmo_record_per_base": (C.c_int, [vp, i64p, u64]),
        "gmo_dummy_cov_nodes": (C.c_long, [vp, i64p, u64, i64p, C.c_long]),
        "gmo_traverse": (C.c_long, [vp, u64, i64p, u64, u64, i64p, C.c_long]),
        "gmo_record_loci": (C.c_int, [vp, i64p, u64]),
        "gmo_num_sites": (u64, [vp]),
        "gmo_is_nested": (C.c_int, [vp]),
        "gmo_allele_sum": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_grouped": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_per_base_nodes": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_node_coverage_at": (C.c_long, [vp, u64, i64p, C.c_long]),
        "gmo_allele_base_non_nested": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_depth_stats": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p, u64p]),
        "gmo_random_access": (None, [vp, i64p]),
        "gmo_target_map": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_par_map": (C.c_long, [vp, i64p, C.c_long]),
        "gmo_bubble_order": (C.c_long, [vp, i64p, C.c_long]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def pack_states(states):
    """states: list of (lo, hi, [(site, allele)...traversed], [(site, allele)...traversing])."""
    v = [len(states)]
    for s in states:
        lo, hi = s[0], s[1]
        tvd = s[2] if len(s) > 2 else []
        tvg = s[3] if len(s) > 3 else []
        v += [lo, hi, len(tvd)]
        for m, a in tvd:
            v += [m, a]
        v.append(len(tvg))
        for m, a in tvg:
            v += [m, a]
    return np.asarray(v, dtype=np.int64)


def unpack_states(v):
    v = [int(x) for x in v]
    n, i, out = v[0], 1, []
    for _ in range(n):
        lo, hi, nt = v[i], v[i + 1], v[i + 2]
        i += 3
        tvd = [(v[i + 2 * j], v[i + 2 * j + 1]) for j in range(nt)]
        i += 2 * nt
        ng = v[i]
        i += 1
        tvg = [(v[i + 2 * j], v[i + 2 * j + 1]) for j in range(ng)]
        i += 2 * ng
        out.append((lo, hi, tvd, tvg))
    return out


class Oracle:
    """One PRG + k-mer index + coverage accumulators, reference semantics."""

    def __init__(self, prg_ints, kmer_size=0, all_kmers=True, rng_mode=RNG_LEMIRE):
        self.lib = _load()
        prg = np.ascontiguousarray(prg_ints, dtype=np.uint32)
        self.prg = prg
        self.k = int(kmer_size)
        self.h = self.lib.gmo_create(_p(prg, C.c_uint32), prg.size, self.k, 1 if all_kmers else 0, rng_mode)
        if not self.h:
            raise RuntimeError(self.lib.gmo_create_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmo_destroy(self.h)
            self.h = None

    __del__ = close

    # -- helpers -----------------------------------------------------------
    def _err(self):
        return RuntimeError(self.lib.gmo_last_error(self.h).decode())

    def _call_out(self, fn, *args, cap=1 << 12):
        while True:
            out = np.empty(cap, dtype=np.int64)
            n = fn(self.h, *args, _p(out, C.c_int64), cap)
            if n == -1:
                raise self._err()
            if n < 0:
                cap = -n
                continue
            return out[:n]

    # -- index -------------------------------------------------------------
    def text_size(self):
        return int(self.lib.gmo_text_size(self.h))

    def sa(self):
        out = np.empty(self.text_size(), dtype=np.uint32)
        self.lib.gmo_sa(self.h, _p(out, C.c_uint32))
        return out

    def bwt(self):
        out = np.empty(self.text_size(), dtype=np.uint32)
        self.lib.gmo_bwt(self.h, _p(out, C.c_uint32))
        return out

    def rank(self, upper, base):
        return int(self.lib.gmo_rank(self.h, upper, base))

    def C_of(self, symbol):
        return int(self.lib.gmo_C_of(self.h, symbol))

    def marker_sa_interval(self, marker):
        lo, hi = C.c_uint32(), C.c_uint32()
        self.lib.gmo_marker_sa_interval(self.h, marker, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def base_next_sa_interval(self, next_char, first_sa, lo, hi):
        a, b = C.c_uint32(), C.c_uint32()
        self.lib.gmo_base_next_sa_interval(self.h, next_char, first_sa, lo, hi, C.byref(a), C.byref(b))
        return a.value, b.value

    def index_kmers(self, kmers):
        arr = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.uint8) for k in kmers]))
        if self.lib.gmo_index_kmers(self.h, _p(arr, C.c_uint8), len(kmers)) != 0:
            raise self._err()

    def index_kmer_diffs(self, diffs):
        """Replace the k-mer index by index_kmers(prefix diffs) (build.cpp:101-131)."""
        flat = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.uint8) for d in diffs]))
        lens = np.asarray([len(d) for d in diffs], dtype=np.uint64)
        if self.lib.gmo_index_kmer_diffs(self.h, _p(flat, C.c_uint8), _p(lens, C.c_uint64), len(diffs)) != 0:
            raise self._err()

    def index_kmers_of_reads(self, reads2d):
        """Index exactly the k-mers that occur in `reads2d` (uint8 [n, L], clean bases) or their reverse complements,
        through the reference's prefix-diff construction (kmers.cpp:42-73 order, build.cpp:101-131). For these reads the
        result equals the all-4^k index — a k-mer without search states is not stored either way — but k = 14 does not
        need 268 M k-mer objects. Test infrastructure, as everything under oracle/."""
        k = self.k
        r = np.ascontiguousarray(reads2d, dtype=np.uint8)
        n, L = r.shape
        codes = []
        for arr in (r, np.ascontiguousarray((5 - r)[:, ::-1])):
            c = np.zeros((n, L - k + 1), dtype=np.uint64)
            for j in range(k):  # base j of the k-mer has weight 4^j: sorting the codes = the reversed-k-mer order
                c |= (arr[:, j:j + L - k + 1].astype(np.uint64) - 1) << np.uint64(2 * j)
            codes.append(np.unique(c))
        self._index_kmer_codes(np.unique(np.concatenate(codes)))

    def _index_kmer_codes(self, u):
        """Sorted unique k-mer codes (base j has weight 4^j) -> prefix diffs -> gmo_index_kmer_diffs."""
        k = self.k
        if u.size == 0:
            return
        km = np.empty((u.size, k), dtype=np.uint8)
        for j in range(k):
            km[:, j] = ((u >> np.uint64(2 * j)) & np.uint64(3)).astype(np.uint8) + 1
        lens = np.full(u.size, k, dtype=np.int64)
        if u.size > 1:
            neq = km[1:] != km[:-1]
            lens[1:] = k - np.argmax(neq[:, ::-1], axis=1)          # up to the highest differing base (kmers.cpp:55-66)
        starts = np.cumsum(lens) - lens
        owner = np.repeat(np.arange(u.size), lens)
        within = np.arange(int(lens.sum())) - starts[owner]
        flat = np.ascontiguousarray(km[owner, within])
        ln = lens.astype(np.uint64)
        if self.lib.gmo_index_kmer_diffs(self.h, _p(flat, C.c_uint8), _p(ln, C.c_uint64), int(u.size)) != 0:
            raise self._err()

    def index_kmers_of_read_list(self, flat, offs):
        """index_kmers_of_reads for ragged reads with errors and Ns (flat uint8, offsets): exactly the k-mers of the clean
        windows of these reads and of their reverse complements — the only k-mers quasimap ever looks up for them
        (quasimap.cpp:206-225). Test infrastructure."""
        k = self.k
        f = np.ascontiguousarray(flat, dtype=np.uint8)
        o = np.asarray(offs, dtype=np.int64)
        n_win = f.size - k + 1
        if n_win <= 0:
            return
        read_of = np.repeat(np.arange(o.size - 1), np.diff(o))
        ok = (f >= 1) & (f <= 4)
        bad_before = np.concatenate([[0], np.cumsum(~ok)])
        valid = (read_of[:n_win] == read_of[k - 1:]) & (bad_before[k:] == bad_before[:n_win])
        codes = []
        for arr in (f, np.ascontiguousarray((5 - f.astype(np.int16))[::-1]).astype(np.uint8)):
            v = valid if arr is f else valid[::-1]
            c = np.zeros(n_win, dtype=np.uint64)
            for j in range(k):
                c |= ((arr[j:j + n_win].astype(np.uint64) - np.uint64(1)) & np.uint64(3)) << np.uint64(2 * j)
            codes.append(np.unique(c[v]))
        u = np.unique(np.concatenate(codes))
        self._index_kmer_codes(u)

    @staticmethod
    def all_kmers(k):
        out = np.empty((4 ** k, k), dtype=np.uint8)
        _load().gmo_all_kmers(k, _p(out, C.c_uint8))
        return out

    @staticmethod
    def prefix_diffs(kmers):
        km = np.ascontiguousarray(kmers, dtype=np.uint8)
        n, k = km.shape
        flat = np.empty(n * k, dtype=np.uint8)
        lens = np.empty(n, dtype=np.uint64)
        _load().gmo_prefix_diffs(_p(km, C.c_uint8), n, k, _p(flat, C.c_uint8), _p(lens, C.c_uint64))
        out, o = [], 0
        for l in lens:
            out.append(flat[o:o + int(l)].tolist())
            o += int(l)
        return out

    @staticmethod
    def reverse_complement(read):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        out = np.empty_like(r)
        _load().gmo_reverse_complement(_p(r, C.c_uint8), r.size, _p(out, C.c_uint8))
        return out

    def all_kmers_in_index(self, read):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        return bool(self.lib.gmo_all_kmers_in_index(self.h, _p(r, C.c_uint8), r.size))

    def max_cov_haplogroup(self, site_index):
        a, c = C.c_int64(), C.c_int64()
        self.lib.gmo_max_cov_haplogroup(self.h, site_index, C.byref(a), C.byref(c))
        return a.value, c.value

    def set_grouped(self, site_index, ids, count):
        v = np.asarray(ids, dtype=np.int64)
        self.lib.gmo_set_grouped(self.h, site_index, _p(v, C.c_int64), v.size, count)

    def kmer_index_size(self):
        return int(self.lib.gmo_kmer_index_size(self.h))

    def kmer_states(self, kmer):
        k = np.ascontiguousarray(kmer, dtype=np.uint8)
        out = self._call_out(self.lib.gmo_kmer_states, _p(k, C.c_uint8))
        if out[0] == -1:
            return None
        return unpack_states(out)

    # -- search ------------------------------------------------------------
    def left_markers_search(self, lo, hi):
        out = self._call_out(self.lib.gmo_left_markers_search, lo, hi)
        return [(int(out[i]), int(out[i + 1])) for i in range(0, len(out), 2)]

    def vbwt_jumps(self, state):
        s = pack_states([state])
        return unpack_states(self._call_out(self.lib.gmo_vbwt_jumps, _p(s, C.c_int64)))

    def search_base_backwards(self, base, states):
        s = pack_states(states)
        return unpack_states(self._call_out(self.lib.gmo_search_base_backwards, base, _p(s, C.c_int64)))

    def process_read_char(self, base, states):
        s = pack_states(states)
        return unpack_states(self._call_out(self.lib.gmo_process_read_char, base, _p(s, C.c_int64)))

    def search_read_backwards(self, read):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        return unpack_states(self._call_out(self.lib.gmo_search_read_backwards, _p(r, C.c_uint8), r.size))

    def encapsulated(self, states):
        s = pack_states(states)
        return unpack_states(self._call_out(self.lib.gmo_encapsulated, _p(s, C.c_int64)))

    def locus_finder(self, state):
        s = pack_states([state])
        out = [int(x) for x in self._call_out(self.lib.gmo_locus_finder, _p(s, C.c_int64))]
        nb = out[0]
        base = out[1:1 + nb]
        nl = out[1 + nb]
        loci = [(out[2 + nb + 2 * j], out[3 + nb + 2 * j]) for j in range(nl)]
        return base, loci

    def extract_max_cov_allele(self, site_marker):
        """(sequence letters, coverage) of the site's most covered allele (read_stats.cpp:94-117)."""
        buf = C.create_string_buffer(1 << 16)
        cov = C.c_int64(0)
        n = self.lib.gmo_extract_max_cov_allele(self.h, site_marker, buf, len(buf), C.byref(cov))
        if n < 0:
            raise self._err()
        return buf.value.decode(), int(cov.value)

    def set_par_map(self, par_map):
        """Mock parental map {site: (parent site, parent allele)} (test_coverage_common.cpp:100-112)."""
        flat = np.asarray([x for k, (ps, pa) in par_map.items() for x in (int(k), ps, pa)] or [0], dtype=np.int64)
        if self.lib.gmo_set_par_map(self.h, _p(flat, C.c_int64), len(par_map)) != 0:
            raise self._err()

    def check_site_uniqueness_throws(self, state):
        r = self.lib.gmo_check_site_uniqueness(self.h, _p(pack_states([state]), C.c_int64))
        if r < 0:
            raise self._err()
        return bool(r)

    def assign_loci(self, loci, traversed_of_states=()):
        """LocusFinder: assign_nested_locus per locus, then assign_traversed_loci per state -> (base, used, loci)."""
        fl = np.asarray([x for l in loci for x in l] or [0], dtype=np.int64)
        st = pack_states(list(traversed_of_states)) if traversed_of_states else None
        out = [int(x) for x in self._call_out(self.lib.gmo_assign_loci, _p(fl, C.c_int64), len(loci),
                                              _p(st, C.c_int64) if st is not None else None)]
        i = 0
        nb = out[i]; base = out[i + 1:i + 1 + nb]; i += 1 + nb
        nu = out[i]; used = out[i + 1:i + 1 + nu]; i += 1 + nu
        nl = out[i]; lo = [(out[i + 1 + 2 * j], out[i + 2 + 2 * j]) for j in range(nl)]
        return base, used, lo

    def unique_site_paths(self, states):
        """MappingInstanceSelector::process_searchstates -> (nonvariant count, [(sites, [(lo, hi)...], loci)...] in map order)."""
        out = [int(x) for x in self._call_out(self.lib.gmo_unique_site_paths, _p(pack_states(states), C.c_int64))]
        nonvar, n, i, entries = out[0], out[1], 2, []
        for _ in range(n):
            ns = out[i]; sites = out[i + 1:i + 1 + ns]; i += 1 + ns
            nst = out[i]; sts = [(out[i + 1 + 2 * j], out[i + 2 + 2 * j]) for j in range(nst)]; i += 1 + 2 * nst
            nl = out[i]; loci = [(out[i + 1 + 2 * j], out[i + 2 + 2 * j]) for j in range(nl)]; i += 1 + 2 * nl
            entries.append((sites, sts, loci))
        return nonvar, entries

    def select_forced(self, states, forced):
        s = pack_states(states)
        out = [int(x) for x in self._call_out(self.lib.gmo_select_forced, _p(s, C.c_int64), forced)]
        called, mn, mx, n_nav, n_loci = out[:5]
        loci = [(out[5 + 2 * j], out[6 + 2 * j]) for j in range(n_loci)]
        return dict(called=called, min=mn, max=mx, n_nav=n_nav, loci=loci)

    # -- rng ---------------------------------------------------------------
    @staticmethod
    def rng_raw(seed, n):
        out = np.empty(n, dtype=np.uint32)
        _load().gmo_rng_raw(seed, n, _p(out, C.c_uint32))
        return out

    @staticmethod
    def rng_generate(seed, lo, hi, n, mode=RNG_LEMIRE):
        out = np.empty(n, dtype=np.uint32)
        _load().gmo_rng_generate(seed, lo, hi, n, mode, _p(out, C.c_uint32))
        return out

    @staticmethod
    def rng_generate_std(seed, lo, hi, n):
        out = np.empty(n, dtype=np.uint32)
        _load().gmo_rng_generate_std(seed, lo, hi, n, _p(out, C.c_uint32))
        return out

    @staticmethod
    def master_seeds(master_seed, reads_per_file):
        rpf = np.ascontiguousarray(reads_per_file, dtype=np.uint64)
        out = np.empty(int(rpf.sum()), dtype=np.uint32)
        _load().gmo_master_seeds(master_seed, _p(rpf, C.c_uint64), rpf.size, _p(out, C.c_uint32))
        return out

    # -- mapping -----------------------------------------------------------
    def quasimap_read(self, read, seed=42):
        r = np.ascontiguousarray(read, dtype=np.uint8)
        if self.lib.gmo_quasimap_read(self.h, _p(r, C.c_uint8), r.size, seed) != 0:
            raise self._err()

    def map_reads(self, reads_flat, offsets, seeds, threads=1):
        """Forward + reverse-complement mapping of every read (quasimap.cpp:82-157)."""
        r = np.ascontiguousarray(reads_flat, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        n = o.size - 1
        if self.lib.gmo_map_reads(self.h, _p(r, C.c_uint8), _p(o, C.c_uint64), _p(s, C.c_uint32), n, threads) != 0:
            raise self._err()

    def stats(self):
        out = np.zeros(5, dtype=np.uint64)
        self.lib.gmo_stats(self.h, _p(out, C.c_uint64))
        return dict(zip(["all", "skipped", "missing_kmer", "no_extension", "exact_mapped"], (int(x) for x in out)))

    def reset_coverage(self):
        self.lib.gmo_reset_coverage(self.h)

    def record_per_base(self, states, read_size):
        s = pack_states(states)
        if self.lib.gmo_record_per_base(self.h, _p(s, C.c_int64), read_size) != 0:
            raise self._err()

    def dummy_cov_nodes(self, states, read_size):
        s = pack_states(states)
        out = [int(x) for x in self._call_out(self.lib.gmo_dummy_cov_nodes, _p(s, C.c_int64), read_size)]
        return {out[1 + 4 * j]: tuple(out[2 + 4 * j:5 + 4 * j]) for j in range(out[0])}

    def traverse(self, prg_pos, path, read_size):
        pp = np.asarray([x for l in path for x in l], dtype=np.int64).reshape(-1)
        if pp.size == 0:
            pp = np.zeros(1, dtype=np.int64)
        out = [int(x) for x in self._call_out(self.lib.gmo_traverse, prg_pos, _p(pp, C.c_int64), len(path), read_size)]
        n = out[0]
        nodes = [dict(node=out[1 + 5 * j], site=out[2 + 5 * j], allele=out[3 + 5 * j],
                      start=out[4 + 5 * j], end=out[5 + 5 * j]) for j in range(n)]
        return nodes, out[1 + 5 * n], (out[2 + 5 * n], out[3 + 5 * n])

    def record_loci(self, loci):
        pp = np.asarray([x for l in loci for x in l], dtype=np.int64)
        if self.lib.gmo_record_loci(self.h, _p(pp, C.c_int64), len(loci)) != 0:
            raise self._err()

    # -- coverage read-back --------------------------------------------------
    def num_sites(self):
        return int(self.lib.gmo_num_sites(self.h))

    def is_nested(self):
        return bool(self.lib.gmo_is_nested(self.h))

    def allele_sum(self):
        out = [int(x) for x in self._call_out(self.lib.gmo_allele_sum, cap=1 << 16)]
        res, i = [], 1
        for _ in range(out[0]):
            n = out[i]
            res.append(out[i + 1:i + 1 + n])
            i += 1 + n
        return res

    def grouped(self):
        """List (per site) of {tuple(allele ids): count}."""
        out = [int(x) for x in self._call_out(self.lib.gmo_grouped, cap=1 << 16)]
        res, i = [], 1
        for _ in range(out[0]):
            ng = out[i]
            i += 1
            d = {}
            for _ in range(ng):
                n = out[i]
                ids = tuple(out[i + 1:i + 1 + n])
                d[ids] = out[i + 1 + n]
                i += 2 + n
            res.append(d)
        return res

    def per_base_nodes(self):
        """List of dict(node, site, allele, first_pos, cov[list]) for every coverage-owning node."""
        out = [int(x) for x in self._call_out(self.lib.gmo_per_base_nodes, cap=1 << 16)]
        res, i = [], 1
        for _ in range(out[0]):
            node, site, allele, fp, n = out[i:i + 5]
            res.append(dict(node=node, site=site, allele=allele, first_pos=fp, cov=out[i + 5:i + 5 + n]))
            i += 5 + n
        return res

    def node_coverage_at(self, pos):
        return [int(x) for x in self._call_out(self.lib.gmo_node_coverage_at, pos)]

    def allele_base_non_nested(self):
        out = [int(x) for x in self._call_out(self.lib.gmo_allele_base_non_nested, cap=1 << 16)]
        res, i = [], 1
        for _ in range(out[0]):
            na = out[i]
            i += 1
            site = []
            for _ in range(na):
                n = out[i]
                site.append(out[i + 1:i + 1 + n])
                i += 1 + n
            res.append(site)
        return res

    def depth_stats(self):
        m, v = C.c_double(), C.c_double()
        a, b = C.c_uint64(), C.c_uint64()
        if self.lib.gmo_depth_stats(self.h, C.byref(m), C.byref(v), C.byref(a), C.byref(b)) != 0:
            raise self._err()
        return dict(mean=m.value, variance=v.value, num_sites_noCov=a.value, num_sites_total=b.value)

    # -- graph introspection ---------------------------------------------------
    def random_access(self):
        n = self.prg.size
        out = np.empty(n * 6, dtype=np.int64)
        self.lib.gmo_random_access(self.h, _p(out, C.c_int64))
        return out.reshape(n, 6)  # node, offset, target_marker, target_allele, node_site, node_allele

    def target_map(self):
        out = [int(x) for x in self._call_out(self.lib.gmo_target_map)]
        res, i = {}, 1
        for _ in range(out[0]):
            key, n = out[i], out[i + 1]
            res[key] = [(out[i + 2 + 2 * j], out[i + 3 + 2 * j]) for j in range(n)]
            i += 2 + 2 * n
        return res

    def par_map(self):
        out = [int(x) for x in self._call_out(self.lib.gmo_par_map)]
        return {out[1 + 3 * j]: (out[2 + 3 * j], out[3 + 3 * j]) for j in range(out[0])}

    def bubble_order(self):
        out = [int(x) for x in self._call_out(self.lib.gmo_bubble_order)]
        return [(out[1 + 3 * j], out[2 + 3 * j], out[3 + 3 * j]) for j in range(out[0])]
