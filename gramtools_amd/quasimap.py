"""Host-side mirror of the reference's quasimap interface over the C-ABI (include/gmx.h).

Names follow libgramtools (paths relative to /root/reference/libgramtools/):
  * ``PRG_Info`` + ``KmerIndex``  -> :class:`Index`              (include/prg/prg_info.hpp:22-59)
  * ``quasimap_reads``            -> :func:`quasimap_reads`      (src/genotype/quasimap/quasimap.cpp:16-57)
  * ``QuasimapReadsStats``        -> :class:`QuasimapReadsStats` (include/genotype/quasimap/quasimap.hpp:17-24)
  * ``Coverage``                  -> :class:`Coverage`           (include/genotype/quasimap/coverage/types.hpp:40-46)
  * ``coverage::dump::*``         -> :func:`dump_allele_sum`, :func:`dump_allele_base`, :func:`dump_grouped_allele_counts`

Mapping always runs on the GPU through libgmx.so; nothing here computes coverage on the CPU.
"""
import ctypes as C
import json
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check

RNG_LEMIRE = 0
GMX_INGEST_BAD_RECORD, GMX_INGEST_BAD_MEMBER, GMX_INGEST_BAD_CRC, GMX_INGEST_TOO_MANY_LINES = 1, 2, 4, 8  # gmx_ingest_result.status
RNG_DIVISION = 1
GROUPED_LOG = 0xFFFFFFFF


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def encode_dna_bases(seq: str) -> np.ndarray:
    """A,C,G,T (any case) -> 1,2,3,4; any other character gives an empty read (common/utils.cpp:73-92)."""
    b = np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]):
        lut[ch] = v
    out = lut[b]
    if out.size and out.min() == 0:
        return np.zeros(0, dtype=np.uint8)
    return out


def master_seeds(master_seed: int, reads_per_file) -> np.ndarray:
    """Per-read selection seeds: 5000 master draws per batch of <= 5000 reads, per file (quasimap.cpp:120-141)."""
    rpf = np.ascontiguousarray(reads_per_file, dtype=np.uint64)
    out = np.empty(int(rpf.sum()), dtype=np.uint32)
    check(_lib.load().gmx_master_seeds(master_seed, _p(rpf, C.c_uint64), rpf.size, _p(out, C.c_uint32)))
    return out


class PinnedArray:
    """A numpy array in page-locked host memory (gmx_host_alloc): what gmx_map_reads_packed_host uploads from
    asynchronously, at the PCIe rate. Freed (returned to the library's cache of page-locked blocks) on close()."""

    def __init__(self, shape, dtype):
        self.lib = _lib.load()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        self.nbytes = max(n * dt.itemsize, 1)
        self.ptr = self.lib.gmx_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("gmx_host_alloc failed")
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.lib.gmx_host_free(self.ptr)
            self.ptr = None

    __del__ = close


class PackedReads:
    """Reads as bit planes in host memory (include/gmx.h, gmx_map_reads_packed_host): ``planes`` (uint64 per 32 bases),
    ``offsets`` (None when ``uniform_len``), ``skip`` (uint8 per read or None), ``n_reads``. ``twobit``: ``planes`` holds
    the reads as a 2-bit stream instead (gmx_map_reads_2bit_host; pack_reads_2bit)."""

    def __init__(self, planes, offsets, uniform_len, skip, n_reads, keep=(), twobit=False):
        self.planes, self.offsets, self.uniform_len, self.skip, self.n_reads = planes, offsets, uniform_len, skip, n_reads
        self.twobit = bool(twobit)
        self._keep = keep  # the PinnedArray objects behind the arrays

    def close(self):
        for k in self._keep:
            k.close()
        self._keep = ()


def pack_reads(reads_flat, offsets, uniform_len: int = 0, threads: int = 0, pinned: bool = False) -> PackedReads:
    """Encoded reads (one byte per base, 1..4) -> bit planes on the host (gmx_pack_reads). With ``uniform_len`` every read
    must have that length and the reads are packed back to back (no offsets needed by the engine). ``pinned``: the
    result lives in page-locked memory, from which the engine uploads asynchronously."""
    lib = _lib.load()
    r = np.ascontiguousarray(reads_flat, dtype=np.uint8)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = o.size - 1
    if r.size == 0:
        r = np.zeros(1, dtype=np.uint8)
    n_pairs = int(lib.gmx_packed_pairs(_p(o, C.c_uint64), uniform_len, n))
    keep = []
    if pinned:
        pa, sk = PinnedArray(n_pairs + 8, np.uint64), PinnedArray(max(n, 1), np.uint8)
        keep = [pa, sk]
        planes, skip = pa.array, sk.array
        planes[n_pairs:] = 0
        off_out = None
        if not uniform_len:
            po = PinnedArray(n + 1, np.uint64)
            keep.append(po)
            po.array[:] = o
            off_out = po.array
    else:
        planes, skip = np.zeros(n_pairs + 8, dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint8)
        off_out = None if uniform_len else o
    check(lib.gmx_pack_reads(r.ctypes.data, o.ctypes.data, uniform_len, n, planes.ctypes.data, skip.ctypes.data, threads))
    return PackedReads(planes, off_out, uniform_len, skip, n, tuple(keep))


def pack_reads_2bit(reads_flat, offsets, uniform_len: int = 0, threads: int = 0, pinned: bool = False) -> PackedReads:
    """Encoded reads -> the 2-bit stream of gmx_map_reads_2bit_host (gmx_pack_reads_2bit): the reads back to back, two bits
    per base, 32 bases per uint64. Returned as a PackedReads whose ``planes`` is the stream and ``twobit`` is True."""
    lib = _lib.load()
    r = np.ascontiguousarray(reads_flat, dtype=np.uint8)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = o.size - 1
    if r.size == 0:
        r = np.zeros(1, dtype=np.uint8)
    units = int(lib.gmx_twobit_units(_p(o, C.c_uint64), uniform_len, n))
    keep = []
    if pinned:
        pa, sk = PinnedArray(units + 8, np.uint64), PinnedArray(max(n, 1), np.uint8)
        keep = [pa, sk]
        stream, skip = pa.array, sk.array
        stream[units:] = 0
        off_out = None
        if not uniform_len:
            po = PinnedArray(n + 1, np.uint64)
            keep.append(po)
            po.array[:] = o
            off_out = po.array
    else:
        stream, skip = np.zeros(units + 8, dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint8)
        off_out = None if uniform_len else o
    check(lib.gmx_pack_reads_2bit(r.ctypes.data, o.ctypes.data, uniform_len, n, stream.ctypes.data, skip.ctypes.data, threads))
    return PackedReads(stream, off_out, uniform_len, skip, n, tuple(keep), twobit=True)


class Index:
    """Everything the mapping path needs, derived from the integer PRG and the k-mer size."""

    def __init__(self, prg, kmer_size: int, threads: int = 0, cache: str = None):
        """`prg`: integer PRG (array) or the path of gram_dir/prg. With `cache` (and a prg path) the index is loaded
        from that cache file when it matches the PRG and k (gmx_index_load), else built."""
        self.lib = _lib.load()
        self.h = C.c_void_p()
        self.from_cache = False
        if cache is not None and isinstance(prg, (str, bytes)):
            path = prg if isinstance(prg, bytes) else prg.encode()
            rc = self.lib.gmx_index_load(cache.encode(), path, kmer_size, C.byref(self.h))
            self.from_cache = rc == 0
            if rc != 0:
                self.h = C.c_void_p()
                check(self.lib.gmx_index_build_from_file(path, kmer_size, threads, C.byref(self.h)))
        elif isinstance(prg, (str, bytes)):
            path = prg if isinstance(prg, bytes) else prg.encode()
            check(self.lib.gmx_index_build_from_file(path, kmer_size, threads, C.byref(self.h)))
        else:
            arr = np.ascontiguousarray(prg, dtype=np.uint32)
            check(self.lib.gmx_index_build(_p(arr, C.c_uint32), arr.size, kmer_size, threads, C.byref(self.h)))
        info = _lib.IndexInfo()
        check(self.lib.gmx_index_get_info(self.h, C.byref(info)))
        self.info = info
        self.kmer_size = info.kmer_size
        self.n_sites = info.n_sites
        self.is_nested = bool(info.is_nested)
        n = self.n_sites
        self.n_alleles = np.zeros(n, dtype=np.uint32)
        self.allele_sum_off = np.zeros(n, dtype=np.uint32)
        self.grouped_off = np.zeros(n, dtype=np.uint32)
        self.parent_site = np.zeros(n, dtype=np.uint32)
        self.parent_allele = np.zeros(n, dtype=np.int32)
        check(self.lib.gmx_index_site_layout(self.h, _p(self.n_alleles, C.c_uint32), _p(self.allele_sum_off, C.c_uint32),
                                             _p(self.grouped_off, C.c_uint32), _p(self.parent_site, C.c_uint32),
                                             _p(self.parent_allele, C.c_int32)))

    @property
    def uses_grouped_log(self):
        """Some site has more than 8 alleles: its grouped counts live in the log, not in dense slots."""
        return bool((self.grouped_off == GROUPED_LOG).any())

    def save(self, path: str):
        """Write the index cache (gmx_index_save)."""
        check(self.lib.gmx_index_save(self.h, path.encode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_index_destroy(self.h)
            self.h = None

    __del__ = close

    # -- layouts -------------------------------------------------------------
    def per_base_layout(self):
        """Rows (site_index, allele, first_prg_pos, per_base_offset, length) of every coverage-owning node."""
        n = self.lib.gmx_index_per_base_layout(self.h, None, 0)
        out = np.zeros((max(n, 1), 5), dtype=np.uint32)
        self.lib.gmx_index_per_base_layout(self.h, _p(out, C.c_uint32), n)
        return out[:n]

    def allele_base_layout(self):
        off = np.zeros(self.info.n_allele_slots, dtype=np.uint32)
        ln = np.zeros(self.info.n_allele_slots, dtype=np.uint32)
        check(self.lib.gmx_index_allele_base_layout(self.h, _p(off, C.c_uint32), _p(ln, C.c_uint32)))
        return off, ln

    # -- introspection (tests) -------------------------------------------------
    def sa(self):
        out = np.empty(self.info.n_text, dtype=np.uint32)
        check(self.lib.gmx_index_copy_sa(self.h, _p(out, C.c_uint32)))
        return out

    def bwt(self):
        out = np.empty(self.info.n_text, dtype=np.uint32)
        check(self.lib.gmx_index_copy_bwt(self.h, _p(out, C.c_uint32)))
        return out

    def rank(self, upper, base):
        return int(self.lib.gmx_index_rank(self.h, upper, base))

    def pos_info(self):
        """Per PRG position: node site, node allele, offset in node, target marker, target allele."""
        n = self.info.n_text - 1
        out = np.empty((n, 5), dtype=np.int64)
        check(self.lib.gmx_index_copy_pos_info(self.h, _p(out, C.c_int64)))
        return out

    def _states_call(self, fn, *args):
        cap = 1 << 12
        while True:
            out = np.empty(cap, dtype=np.int64)
            n = fn(self.h, *args, _p(out, C.c_int64), cap)
            if n < -1:
                cap = -n
                continue
            check(n)
            return [int(x) for x in out[:n]]

    @staticmethod
    def _unpack_states(v):
        if v[0] == -1:
            return None
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

    def target_map(self):
        v = self._states_call(self.lib.gmx_index_copy_target_map)
        res, i = {}, 1
        for _ in range(v[0]):
            key, n = v[i], v[i + 1]
            res[key] = [(v[i + 2 + 2 * j], v[i + 3 + 2 * j]) for j in range(n)]
            i += 2 + 2 * n
        return res

    def seed_states(self, kmer):
        """k-mer index entry (build/kmer_index/build.cpp:101-131) or None when the k-mer is absent. A k-mer of length
        info.kmer_size2 is looked up in the longer seed table."""
        k = np.ascontiguousarray(kmer, dtype=np.uint8)
        return self._unpack_states(self._states_call(self.lib.gmx_index_seed_states_k, _p(k, C.c_uint8), k.size))

    def bubble_order(self):
        out = np.zeros(max(self.n_sites, 1), dtype=np.uint32)
        check(self.lib.gmx_index_bubble_order(self.h, _p(out, C.c_uint32)))
        return out[:self.n_sites].tolist()

    def jump_states(self, lo, hi):
        """search_state_vBWT_jumps of the path-less state [lo, hi] (vBWT_jump.cpp:134-183)."""
        return self._unpack_states(self._states_call(self.lib.gmx_index_jump_states, lo, hi))


@dataclass
class QuasimapReadsStats:
    all_reads_count: int = 0
    skipped_reads_count: int = 0
    missing_kmer_reads_count: int = 0
    no_extension_reads_count: int = 0
    exact_mapped_reads_count: int = 0

    def as_dict(self):
        return dict(all=self.all_reads_count, skipped=self.skipped_reads_count, missing_kmer=self.missing_kmer_reads_count,
                    no_extension=self.no_extension_reads_count, exact_mapped=self.exact_mapped_reads_count)


class Coverage:
    """Final coverage with the reference's uint16 semantics applied to the device totals."""

    def __init__(self, index: Index, allele_sum_u32, per_base_u32, grouped_u32, grouped_log, stats: QuasimapReadsStats):
        self.index = index
        self.raw_allele_sum = allele_sum_u32
        self.raw_per_base = per_base_u32
        self.raw_grouped = grouped_u32
        self.raw_grouped_log = grouped_log
        self.stats = stats
        self.allele_sum_flat = (allele_sum_u32 & 0xFFFF).astype(np.uint16)             # wraps, data_types.hpp:52
        self.per_base_flat = np.minimum(per_base_u32, 65535).astype(np.uint16)           # saturates, allele_base.cpp:239

    # AlleleSumCoverage: vector (per site) of vector (per allele)
    @property
    def allele_sum_coverage(self):
        ix = self.index
        return [[int(x) for x in self.allele_sum_flat[o:o + n]] for o, n in zip(ix.allele_sum_off, ix.n_alleles)]

    # SitesGroupedAlleleCounts: per site {tuple(allele ids): count}
    @property
    def grouped_allele_counts(self):
        ix = self.index
        sites = [dict() for _ in range(ix.n_sites)]
        for s in range(ix.n_sites):
            off = int(ix.grouped_off[s])
            if off == GROUPED_LOG:
                continue
            n = int(ix.n_alleles[s])
            vals = self.raw_grouped[off:off + (1 << n) - 1]
            for m in np.nonzero(vals)[0]:
                c = int(vals[m]) & 0xFFFF  # a total that wrapped to 0 keeps its (zero-valued) key, as the reference's map does
                mask = int(m) + 1
                ids = tuple(a for a in range(n) if (mask >> a) & 1)
                sites[s][ids] = c
        for s, ids, count in iter_grouped_log(self.raw_grouped_log):
            sites[s][ids] = (sites[s].get(ids, 0) + count) & 0xFFFF
        return sites

    # SitesAlleleBaseCoverage (allele_base_non_nested, allele_base.cpp:10-38); [] for nested PRGs
    @property
    def allele_base_coverage(self):
        ix = self.index
        if ix.is_nested:
            return []
        off, ln = ix.allele_base_layout()
        out = []
        for s in range(ix.n_sites):
            site = []
            for a in range(int(ix.n_alleles[s])):
                slot = int(ix.allele_sum_off[s]) + a
                site.append([int(x) for x in self.per_base_flat[off[slot]:off[slot] + ln[slot]]])
            out.append(site)
        return out

    def depth_stats(self):
        """read_stats.json depth block: mean / population variance of per-site max-haplogroup coverage over
        level-0 sites (AbstractReadStats::compute_coverage_depth, read_stats.cpp:119-160)."""
        d = _lib.DepthStats()
        pb = np.ascontiguousarray(self.raw_per_base if self.raw_per_base.size else np.zeros(1, np.uint32), dtype=np.uint32)
        g = np.ascontiguousarray(self.raw_grouped if self.raw_grouped.size else np.zeros(1, np.uint32), dtype=np.uint32)
        lg = np.ascontiguousarray(self.raw_grouped_log if self.raw_grouped_log.size else np.zeros(1, np.uint32), dtype=np.uint32)
        check(_lib.load().gmx_compute_coverage_depth(self.index.h, _p(pb, C.c_uint32), _p(g, C.c_uint32), _p(lg, C.c_uint32),
                                                     self.raw_grouped_log.size, C.byref(d)))
        return dict(mean=d.mean_cov_depth, variance=d.variance_cov_depth, num_sites_noCov=d.num_sites_noCov,
                    num_sites_total=d.num_sites_total)

    def per_base_by_first_pos(self):
        """{first PRG position of a coverage-owning node: [per-base counts]} (works for nested PRGs too)."""
        return {int(r[2]): [int(x) for x in self.per_base_flat[r[3]:r[3] + r[4]]] for r in self.index.per_base_layout()}


def dump_allele_sum(cov: Coverage) -> str:
    """coverage/allele_sum_coverage text (allele_sum.cpp:45-57)."""
    return "".join(" ".join(str(c) for c in site) + "\n" for site in cov.allele_sum_coverage)


def allele_base_json(sites) -> str:
    """dump_allele_base_coverage (allele_base.cpp:49-107): {"allele_base_counts":[[[per-base of allele 0],...],...]}, compact."""
    body = ",".join("[" + ",".join("[" + ",".join(str(int(c)) for c in al) + "]" for al in site) + "]" for site in sites)
    return '{"allele_base_counts":[' + body + "]}"


def dump_allele_base(cov: Coverage) -> str:
    """coverage/allele_base_coverage.json (allele_base.cpp:49-107)."""
    return allele_base_json(cov.allele_base_coverage) + "\n"


def hash_allele_groups(sites) -> dict:
    """hash_allele_groups (grouped_allele_counts.cpp:51-67): every distinct allele-id group of any site gets a group id,
    ids are 0, 1, 2, ... without gaps. The reference hands them out in unordered_map order (labels); here: first
    appearance, sites in order, a site's groups sorted."""
    group_id = {}
    for site in sites:
        for ids in sorted(site):
            group_id.setdefault(tuple(ids), len(group_id))
    return group_id


def group_id_counts(sites, group_hash):
    """get_group_id_counts (grouped_allele_counts.cpp:69-84): per site {group id (as a string): count}, ordered by the numeric id."""
    out = []
    for site in sites:
        d = {group_hash[tuple(ids)]: int(c) for ids, c in site.items()}
        out.append({str(g): d[g] for g in sorted(d)})
    return out


def group_id_alleles(group_hash):
    """get_group_id_alleles (grouped_allele_counts.cpp:86-93): {group id (string): allele ids}, ordered by the NUMERIC id."""
    return {str(g): list(ids) for ids, g in sorted(group_hash.items(), key=lambda kv: kv[1])}


def grouped_json(sites, group_hash) -> str:
    """get_json(...).dump() (grouped_allele_counts.cpp:95-110): compact, keys in the order above."""
    groups = ",".join(f'"{g}":[' + ",".join(str(a) for a in ids) + "]" for g, ids in group_id_alleles(group_hash).items())
    counts = ",".join("{" + ",".join(f'"{g}":{c}' for g, c in d.items()) + "}" for d in group_id_counts(sites, group_hash))
    return '{"grouped_allele_counts":{"allele_groups":{' + groups + '},"site_counts":[' + counts + "]}}"


def dump_grouped_allele_counts(cov: Coverage) -> str:
    """coverage/grouped_allele_counts_coverage.json (grouped_allele_counts.cpp:51-110). Group ids are labels
    (the reference assigns them in unordered_map order); here: first appearance in site order."""
    sites = cov.grouped_allele_counts
    return grouped_json(sites, hash_allele_groups(sites)) + "\n"


LOG_COUNTED = 0x80000000
LOG_PAD = 0xFFFFFFFF


def iter_grouped_log(log):
    """(site_index, ids, count) of every record of a grouped log (gmx.h, gmx_coverage_fetch_grouped_log): records worth
    +1 ``[site, n, ids...]`` or +count ``[site, n | LOG_COUNTED, count_lo, count_hi, ids...]``; the same key may recur."""
    i, n_words = 0, len(log)
    while i < n_words:
        if int(log[i]) == LOG_PAD:
            i += 1
            continue
        s, n = int(log[i]), int(log[i + 1])
        head, count = 2, 1
        if n & LOG_COUNTED:
            n &= ~LOG_COUNTED
            head, count = 4, int(log[i + 2]) | (int(log[i + 3]) << 32)
        yield s, tuple(int(np.int32(x)) for x in log[i + head:i + head + n]), count
        i += head + n


class Quasimapper:
    """An engine on one GPU: the index resident in HBM plus zeroed coverage accumulators."""

    def __init__(self, index: Index, device: int = 0, rng_mode: int = RNG_LEMIRE, max_states: int = 0,
                 max_path_nodes: int = 0, max_batch_reads: int = 0, forward_only: bool = False,
                 huge_heap_bytes: int = 0, log_cap_words: int = 0):
        self.lib = _lib.load()
        self.index = index
        opts = _lib.EngineOpts()
        self.lib.gmx_engine_default_opts(C.byref(opts))
        opts.device = device
        opts.rng_mode = rng_mode
        if max_states:
            opts.max_states = max_states
        if max_path_nodes:
            opts.max_path_nodes = max_path_nodes
        if max_batch_reads:
            opts.max_batch_reads = max_batch_reads
        opts.forward_only = 1 if forward_only else 0
        if huge_heap_bytes:
            opts.huge_heap_bytes = huge_heap_bytes
        if log_cap_words:
            opts.log_cap_words = log_cap_words
        self.h = C.c_void_p()
        check(self.lib.gmx_engine_create(index.h, C.byref(opts), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, stream=None):
        """Zero coverage and statistics; with `stream` (a hipStream_t value) the memsets are enqueued, not awaited."""
        if stream is None:
            check(self.lib.gmx_engine_reset(self.h))
        else:
            check(self.lib.gmx_engine_reset_async(self.h, C.c_void_p(stream) if stream else None))

    def map_reads(self, reads_flat, offsets, seeds):
        """Host buffers: forward + reverse-complement mapping of every read (quasimap.cpp:82-157)."""
        r = np.ascontiguousarray(reads_flat, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        if r.size == 0:
            r = np.zeros(1, dtype=np.uint8)
        check(self.lib.gmx_map_reads_host(self.h, _p(r, C.c_uint8), _p(o, C.c_uint64), _p(s, C.c_uint32), o.size - 1))

    def map_reads_packed(self, packed: "PackedReads", seeds, use_skip=True):
        """Reads already packed to bit planes on the host (gmx_map_reads_packed_host). Asynchronous when the arrays are
        page-locked (pack_reads(..., pinned=True), PinnedArray): keep them untouched until sync_uploads() / sync()."""
        s = seeds if isinstance(seeds, np.ndarray) and seeds.dtype == np.uint32 and seeds.flags.c_contiguous else \
            np.ascontiguousarray(seeds, dtype=np.uint32)
        skip = packed.skip if use_skip else None
        fn = self.lib.gmx_map_reads_2bit_host if packed.twobit else self.lib.gmx_map_reads_packed_host
        check(fn(self.h, packed.planes.ctypes.data, None if packed.offsets is None else packed.offsets.ctypes.data,
                 packed.uniform_len, s.ctypes.data, None if skip is None else skip.ctypes.data, packed.n_reads))
        self._last_seeds = s  # (kept alive while the upload may be in flight)

    def sync_uploads(self):
        check(self.lib.gmx_engine_sync_uploads(self.h))

    def seeds_in_place(self, on: bool = True):
        """map_reads_packed uploads no seeds when they lie in page-locked memory: the kernels read the few they need from
        there (gmx_engine_seeds_in_place). The seeds must then stay untouched until sync() / coverage()."""
        check(self.lib.gmx_engine_seeds_in_place(self.h, 1 if on else 0))

    def map_ingested(self, res, seeds: "PinnedArray", first: int = 0):
        """The reads a slot of an Ingest holds (gmx_ingest_result), mapped where they lie in HBM (gmx_map_reads_packed_device);
        `seeds`: a PinnedArray of uint32 read in place by the kernels, the chunk's first read taking seeds[first]."""
        n = int(res.n_reads)
        sp = seeds.ptr + 4 * first
        if res.uniform_len:
            check(self.lib.gmx_map_reads_packed_device(self.h, res.d_planes, None, res.uniform_len, sp, res.d_skip if res.any_skip else None, n))
        else:
            for i, r0 in enumerate(range(0, n, 1 << 20)):
                m = min(1 << 20, n - r0)
                check(self.lib.gmx_map_reads_packed_device(self.h, res.d_planes + 8 * int(res.sub_pairs[i]), res.d_offsets + 8 * r0, 0,
                                                           sp + 4 * r0, (res.d_skip + r0) if res.any_skip else None, m))
        self._last_seeds = seeds

    def map_reads_device(self, d_reads, d_offsets, d_seeds, n_reads, stream=None):
        """Device-resident buffers (torch CUDA tensors: uint8 / int64-or-uint64 / int32-or-uint32). Asynchronous."""
        sp = C.c_void_p(stream) if stream else None
        check(self.lib.gmx_map_reads_device(self.h, C.c_void_p(d_reads.data_ptr()), C.c_void_p(d_offsets.data_ptr()),
                                            C.c_void_p(d_seeds.data_ptr()), n_reads, d_reads.numel(), sp))

    def sync(self):
        check(self.lib.gmx_engine_sync(self.h))

    def enable_timing(self, on=True):
        check(self.lib.gmx_engine_enable_timing(self.h, 1 if on else 0))

    def timing(self):
        """HIP-event time of the kernels since the last call: dict(search_ms, search_launches, cover_ms, ..., reads)."""
        t = _lib.Timing()
        check(self.lib.gmx_engine_timing(self.h, C.byref(t)))
        names = ("seed", "filter0", "filter1", "single", "extend2", "unpack")
        kernels = {nm: dict(ms=t.kernel_ms[i], launches=int(t.kernel_launches[i])) for i, nm in enumerate(names)}
        return dict(search_ms=t.search_ms, search_launches=t.search_launches, cover_ms=t.cover_ms,
                    cover_launches=t.cover_launches, reads=t.reads, kernels=kernels)

    def device_coverage(self):
        dc = _lib.DeviceCoverage()
        check(self.lib.gmx_coverage_device(self.h, C.byref(dc)))
        return dc

    def queue_counts(self):
        """Queue lengths of the last batch (which route the tasks took); see gmx_queue_counts."""
        q = _lib.QueueCounts()
        check(self.lib.gmx_engine_queue_counts(self.h, C.byref(q)))
        return {n: int(getattr(q, n)) for n, _ in q._fields_}

    # ---- test hooks: SearchStates of the HIP path (gmx.h; states as (lo, hi, [(site, allele)...], [(site, -1)...])) ----
    @staticmethod
    def _states_to_words(states):
        w = [len(states)]
        for lo, hi, tvd, tvg in states:
            w += [lo, hi, len(tvd), len(tvg)]
            for site, allele in tvd:
                w += [site, allele & 0xFFFFFFFF]
            w += [x[0] if isinstance(x, (tuple, list)) else x for x in tvg]
        return np.asarray(w, dtype=np.uint32)

    @staticmethod
    def _words_to_states(w):
        out, at = [], 1
        for _ in range(int(w[0])):
            lo, hi, nt, ng = (int(x) for x in w[at:at + 4])
            at += 4
            tvd = [(int(w[at + 2 * j]), int(np.int32(w[at + 2 * j + 1]))) for j in range(nt)]
            at += 2 * nt
            tvg = [(int(w[at + j]), -1) for j in range(ng)]
            at += ng
            out.append((lo, hi, tvd, tvg))
        return out

    def debug_keep_states(self, on=True):
        """From the next batch on every task's final states stay readable (debug_final_states)."""
        check(self.lib.gmx_engine_debug_keep_states(self.h, 1 if on else 0))

    def debug_final_states(self, read, orientation=0):
        """Final SearchStates of (read, orientation) of the last launch, as search_read_backwards leaves them. -> (states, tier)"""
        n, tier = C.c_uint64(0), C.c_int(-1)
        check(self.lib.gmx_debug_final_states(self.h, 2 * read + orientation, None, 0, C.byref(n), C.byref(tier)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint32)
        check(self.lib.gmx_debug_final_states(self.h, 2 * read + orientation, _p(buf, C.c_uint32), buf.size, C.byref(n), C.byref(tier)))
        return self._words_to_states(buf), tier.value

    def debug_search(self, read, states=None, from_pos=None, stop=0, lf_only=False):
        """The device's search loop on one read. states=None: seeded from the k-mer index (search_read_backwards); else from
        the given states at read position from_pos (default: the whole read is still to be matched) down to stop."""
        r = np.ascontiguousarray(read, dtype=np.uint8)
        seeded = states is None
        sw = np.zeros(1, dtype=np.uint32) if seeded else self._states_to_words(states)
        frm = r.size if from_pos is None else from_pos
        n = C.c_uint64(0)
        cap = 1 << 16
        while True:
            buf = np.zeros(cap, dtype=np.uint32)
            rc = self.lib.gmx_debug_search(self.h, _p(r, C.c_uint8), r.size, 1 if seeded else 0, _p(sw, C.c_uint32), sw.size, frm, stop,
                                           1 if lf_only else 0, _p(buf, C.c_uint32), buf.size, C.byref(n))
            if rc == -4 and n.value > cap:
                cap = int(n.value)
                continue
            check(rc)
            return self._words_to_states(buf)

    def debug_encapsulate(self, states):
        """The device's handle_allele_encapsulated_states on the given states -> (states inside sites, SA indices outside sites)."""
        sw = self._states_to_words(states)
        positions = sum(hi - lo + 1 for lo, hi, _, _ in states) + len(states) + 1
        buf = np.zeros(8 * positions + 64 + sw.size, dtype=np.uint32)
        nv = np.zeros(positions, dtype=np.uint32)
        n, n_nv = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.gmx_debug_encapsulate(self.h, _p(sw, C.c_uint32), sw.size, _p(buf, C.c_uint32), buf.size, C.byref(n),
                                             _p(nv, C.c_uint32), nv.size, C.byref(n_nv)))
        return self._words_to_states(buf), [int(x) for x in nv[:n_nv.value]]

    def reduce_begin(self, stream=None):
        """Before the all-reduce of the fused block: read counters -> 16-bit limbs inside it."""
        check(self.lib.gmx_coverage_reduce_begin(self.h, C.c_void_p(stream) if stream else None))

    def reduce_end(self, stream=None):
        """After the all-reduce: limb sums -> read counters."""
        check(self.lib.gmx_coverage_reduce_end(self.h, C.c_void_p(stream) if stream else None))

    def import_grouped_log(self, records, replace=False):
        """Adds (or, with `replace`, substitutes) grouped-log records to this engine's totals (log exchanges done in Python)."""
        r = np.ascontiguousarray(records, dtype=np.uint32)
        check(self.lib.gmx_coverage_import_grouped_log(self.h, _p(r if r.size else np.zeros(1, np.uint32), C.c_uint32), r.size,
                                                       1 if replace else 0))

    def grouped_log(self):
        n = check(self.lib.gmx_coverage_fetch_grouped_log(self.h, None, 0))
        log = np.zeros(max(n, 1), dtype=np.uint32)
        if n:
            check(self.lib.gmx_coverage_fetch_grouped_log(self.h, _p(log, C.c_uint32), n))
        return log[:n]

    def coverage(self) -> Coverage:
        info = self.index.info
        a = np.zeros(max(info.n_allele_slots, 1), dtype=np.uint32)
        p = np.zeros(max(info.n_per_base_slots, 1), dtype=np.uint32)
        g = np.zeros(max(info.n_grouped_slots, 1), dtype=np.uint32)
        st = _lib.Stats()
        check(self.lib.gmx_coverage_fetch(self.h, _p(a, C.c_uint32), _p(p, C.c_uint32), _p(g, C.c_uint32), C.byref(st)))
        n = check(self.lib.gmx_coverage_fetch_grouped_log(self.h, None, 0))
        log = np.zeros(max(n, 1), dtype=np.uint32)
        if n:
            check(self.lib.gmx_coverage_fetch_grouped_log(self.h, _p(log, C.c_uint32), n))
        stats = QuasimapReadsStats(st.all_reads_count, st.skipped_reads_count, st.missing_kmer_reads_count,
                                   st.no_extension_reads_count, st.exact_mapped_reads_count)
        return Coverage(self.index, a[:info.n_allele_slots], p[:info.n_per_base_slots], g[:info.n_grouped_slots], log[:n], stats)


def bgzf_members(data) -> list:
    """The members of a BGZF file (SAM spec 4.1) as (offset of the deflate data, its size, isize, crc32) without inflating
    anything: the walk `gram` does before it hands a file to the device (gmx_ingest_submit_bgzf). Raises ValueError on
    anything that is not a BGZF member; the EOF marker (an empty member) is dropped."""
    import struct
    mv = memoryview(data)
    out, at, n = [], 0, len(mv)
    while at < n:
        if at + 18 > n or bytes(mv[at:at + 4]) != b"\x1f\x8b\x08\x04":
            raise ValueError(f"no BGZF member at byte {at}")
        xlen = struct.unpack_from("<H", mv, at + 10)[0]
        x, bsize = 0, None
        while x + 4 <= xlen:
            si1, si2, slen = struct.unpack_from("<BBH", mv, at + 12 + x)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", mv, at + 12 + x + 4)[0] + 1
            x += 4 + slen
        if bsize is None or bsize < 12 + xlen + 8 or at + bsize > n:
            raise ValueError(f"damaged BGZF member at byte {at}")
        crc, isize = struct.unpack_from("<II", mv, at + bsize - 8)
        if isize:
            out.append((at + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
        at += bsize
    return out


class Ingest:
    """Reads files decoded on the device (include/gmx.h, gmx_ingest_*): BGZF members inflated, four-line records found and
    packed into bit planes by HIP kernels; three slots (0, 1, 2) taken in turn. ``submit_bgzf`` / ``submit_text`` enqueue a chunk, ``wait`` returns
    its gmx_ingest_result; Quasimapper.map_ingested maps what a slot holds."""

    def __init__(self, device: int = 0, max_text_bytes: int = 64 << 20):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        check(self.lib.gmx_ingest_create(device, max_text_bytes, C.byref(self.h)))
        self._keep = [None, None, None]

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_ingest_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        check(self.lib.gmx_ingest_reset(self.h))

    @staticmethod
    def member_array(members):
        """(offset, size, isize, crc32) tuples -> the gmx_bgzf_member array gmx_ingest_submit_bgzf takes (a Python loop: ~0.4 us a member)."""
        arr = (_lib.BgzfMember * max(len(members), 1))()
        for i, (off, size, isize, crc) in enumerate(members):
            arr[i] = _lib.BgzfMember(off, size, isize, crc, 0)
        arr.n_members = len(members)
        return arr

    def submit_bgzf(self, slot: int, data, members, final: bool):
        """`data`: bytes-like holding the chunk's members; `members`: (offset, size, isize, crc32) of each, offsets into data."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        arr = members if isinstance(members, C.Array) else self.member_array(members)  # (a caller in a hurry builds the array ahead)
        self._keep[slot] = (buf, arr)
        check(self.lib.gmx_ingest_submit_bgzf(self.h, slot, buf.ctypes.data if buf.size else None, buf.size, arr, getattr(arr, 'n_members', len(members)), 1 if final else 0))

    def submit_bgzf_deferred(self, slot: int, data, members):
        """Upload + inflate only (chunks dealt over several devices): gmx_ingest_scan follows when the end of the chunk before is known."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        arr = (_lib.BgzfMember * max(len(members), 1))()
        for i, (off, size, isize, crc) in enumerate(members):
            arr[i] = _lib.BgzfMember(off, size, isize, crc, 0)
        self._keep[slot] = (buf, arr)
        check(self.lib.gmx_ingest_submit_bgzf_deferred(self.h, slot, buf.ctypes.data if buf.size else None, buf.size, arr, len(members)))

    def submit_text_deferred(self, slot: int, text):
        """Upload only (plain text dealt over several devices): gmx_ingest_scan follows when the end of the chunk before is known."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        self._keep[slot] = (buf,)
        check(self.lib.gmx_ingest_submit_text_deferred(self.h, slot, buf.ctypes.data if buf.size else None, buf.size))

    def scan(self, slot: int, carry: bytes, final: bool):
        c = np.frombuffer(carry, dtype=np.uint8) if carry else np.zeros(0, dtype=np.uint8)
        check(self.lib.gmx_ingest_scan(self.h, slot, c.ctypes.data if c.size else None, c.size, 1 if final else 0))

    def fetch_tail(self, slot: int) -> bytes:
        n = self.lib.gmx_ingest_fetch_tail(self.h, slot, None, 0)
        if n < 0:
            check(int(n))
        out = np.zeros(max(int(n), 1), dtype=np.uint8)
        got = self.lib.gmx_ingest_fetch_tail(self.h, slot, out.ctypes.data, out.size)
        if got < 0:
            check(int(got))
        return out[:int(n)].tobytes()

    def submit_text(self, slot: int, text, final: bool):
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        self._keep[slot] = (buf,)
        check(self.lib.gmx_ingest_submit_text(self.h, slot, buf.ctypes.data if buf.size else None, buf.size, 1 if final else 0))

    def wait(self, slot: int) -> "_lib.IngestResult":
        res = _lib.IngestResult()
        check(self.lib.gmx_ingest_wait(self.h, slot, C.byref(res)))
        self._keep[slot] = None
        return res

    def release_after(self, slot: int, stream=None, engine=None):
        """The slot's next submit waits for the work enqueued on `stream` (default: the NULL stream); with `engine` (the Quasimapper
        that just mapped the slot's reads) also for the launches on its second workspace's stream (gmx_engine_second_stream)."""
        check(self.lib.gmx_ingest_release_after(self.h, slot, C.c_void_p(stream) if stream else None))
        if engine is not None:
            second = self.lib.gmx_engine_second_stream(engine.h)
            if second:
                check(self.lib.gmx_ingest_release_after(self.h, slot, C.c_void_p(second)))

    def fetch_text(self, slot: int) -> bytes:
        n = self.lib.gmx_ingest_fetch_text(self.h, slot, None, 0)
        if n < 0:
            check(int(n))
        out = np.zeros(max(int(n), 1), dtype=np.uint8)
        got = self.lib.gmx_ingest_fetch_text(self.h, slot, out.ctypes.data, out.size)
        if got < 0:
            check(int(got))
        return out[:int(n)].tobytes()

    def fetch_reads(self, slot: int, res) -> "PackedReads":
        """The slot's reads in the host layout (bit planes, offsets, skip flags): what the host parser produces of the same text."""
        planes = np.zeros(int(res.n_pairs) + 8, dtype=np.uint64)
        offsets = None if res.uniform_len else np.zeros(int(res.n_reads) + 1, dtype=np.uint64)
        skip = np.zeros(max(int(res.n_reads), 1), dtype=np.uint8)
        check(self.lib.gmx_ingest_fetch_reads(self.h, slot, planes.ctypes.data, None if offsets is None else offsets.ctypes.data, skip.ctypes.data))
        return PackedReads(planes, offsets, int(res.uniform_len), skip, int(res.n_reads))


class QuasimapperGroup:
    """Several GPUs of one node in one process (gmx.h: gmx_group_*): an engine per listed device, reads dealt by
    read index, one exchange at the end. ``devices`` may repeat an ordinal (two engines on one GPU: the exchange then
    runs over peer copies instead of RCCL) — which is how the path is tested on a one-GPU box."""

    def __init__(self, index: Index, devices, rng_mode: int = RNG_LEMIRE, **opts_kw):
        self.lib = _lib.load()
        self.index = index
        opts = _lib.EngineOpts()
        self.lib.gmx_engine_default_opts(C.byref(opts))
        opts.rng_mode = rng_mode
        for k, v in opts_kw.items():
            setattr(opts, k, v)
        dev = (C.c_int * len(devices))(*devices)
        self.h = C.c_void_p()
        check(self.lib.gmx_group_create(index.h, C.byref(opts), dev, len(devices), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_group_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def uses_rccl(self):
        return bool(self.lib.gmx_group_uses_rccl(self.h))

    def map_reads(self, reads_flat, offsets, seeds):
        r = np.ascontiguousarray(reads_flat, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        if r.size == 0:
            r = np.zeros(1, dtype=np.uint8)
        check(self.lib.gmx_group_map_reads_host(self.h, _p(r, C.c_uint8), _p(o, C.c_uint64), _p(s, C.c_uint32), o.size - 1))

    def map_reads_packed(self, packed: "PackedReads", seeds, use_skip=True):
        if packed.twobit:  # a member's share of a 2-bit stream does not start on a unit boundary: the group takes planes
            raise ValueError("QuasimapperGroup.map_reads_packed takes bit planes (pack_reads), not a 2-bit stream (pack_reads_2bit)")
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        skip = packed.skip if use_skip else None
        check(self.lib.gmx_group_map_reads_packed_host(
            self.h, packed.planes.ctypes.data, None if packed.offsets is None else packed.offsets.ctypes.data,
            packed.uniform_len, s.ctypes.data, None if skip is None else skip.ctypes.data, packed.n_reads))
        self._last_seeds = s
        check(self.lib.gmx_group_sync_uploads(self.h))

    def allreduce(self):
        check(self.lib.gmx_group_allreduce(self.h))

    def coverage(self, member: int = 0) -> Coverage:
        """Coverage held by engine `member` (after :meth:`allreduce`: the totals of the whole job, on every member)."""
        view = Quasimapper.__new__(Quasimapper)
        view.lib, view.index, view.h = self.lib, self.index, C.c_void_p(self.lib.gmx_group_engine(self.h, member))
        try:
            return view.coverage()
        finally:
            view.h = None  # the group owns the engine


class Genotyped:
    """The infer stage (gmx.h: gmx_infer_*; LevelGenotyper + writers, src/genotype/genotype.cpp:72-118) on a Coverage:
    level genotyping of every site from the recorded coverage. Host work."""

    def __init__(self, cov: Coverage, mean_pb_error: float, ploidy: str = "haploid", depth=None):
        import json as _json
        self._json = _json
        self.lib = _lib.load()
        self.index = cov.index
        d = depth if depth is not None else cov.depth_stats()
        pb = np.ascontiguousarray(cov.raw_per_base if cov.raw_per_base.size else np.zeros(1, np.uint32), dtype=np.uint32)
        g = np.ascontiguousarray(cov.raw_grouped if cov.raw_grouped.size else np.zeros(1, np.uint32), dtype=np.uint32)
        lg = np.ascontiguousarray(cov.raw_grouped_log if cov.raw_grouped_log.size else np.zeros(1, np.uint32), dtype=np.uint32)
        self.h = C.c_void_p()
        check(self.lib.gmx_infer_run(self.index.h, _p(pb, C.c_uint32), _p(g, C.c_uint32), _p(lg, C.c_uint32),
                                     cov.raw_grouped_log.size, d["mean"], d["variance"], mean_pb_error,
                                     1 if ploidy == "haploid" else 2, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmx_infer_destroy(self.h)
            self.h = None

    __del__ = close

    def site(self, i: int) -> dict:
        n = check(self.lib.gmx_infer_site_json(self.h, i, None, 0))
        buf = C.create_string_buffer(n + 1)
        check(self.lib.gmx_infer_site_json(self.h, i, buf, n + 1))
        return self._json.loads(buf.value.decode())

    def called_alleles(self, i: int):
        """Sequences of the distinct genotyped alleles of site i ([] when null genotyped)."""
        s = self.site(i)
        if s["GT"][0][0] is None:
            return []
        return [s["ALS"][g] for g in sorted(set(s["GT"][0]))]

    def write(self, genotype_dir: str, sample_id: str, coords_path: str = None):
        import os
        cp = coords_path.encode() if coords_path else None
        check(self.lib.gmx_infer_write_json(self.h, cp, sample_id.encode(), os.path.join(genotype_dir, "genotyped.json").encode()))
        check(self.lib.gmx_infer_write_vcf(self.h, cp, sample_id.encode(), os.path.join(genotype_dir, "genotyped.vcf.gz").encode()))
        desc = f"{sample_id} personalised reference made by gramtools genotype"
        check(self.lib.gmx_infer_write_fasta(self.h, cp, desc.encode(),
                                             os.path.join(genotype_dir, "personalised_reference.fasta").encode()))


def genotyping_model(alleles, grouped_counts, ploidy, mean_cov, var_cov, mean_pb_error) -> dict:
    """LevelGenotyperModel on explicit data (gmx_infer_model): alleles = [(sequence, [per-base cov], haplogroup, callable)],
    grouped_counts = {(ids...): count}."""
    import json as _json
    lib = _lib.load()
    n = len(alleles)
    seqs = (C.c_char_p * n)(*[a[0].encode() for a in alleles])
    pb_off = np.concatenate([[0], np.cumsum([len(a[1]) for a in alleles])]).astype(np.uint32)
    pb = np.asarray([c for a in alleles for c in a[1]] or [0], dtype=np.uint32)
    hap = np.asarray([a[2] for a in alleles], dtype=np.int32)
    call = np.asarray([1 if (len(a) < 4 or a[3]) else 0 for a in alleles], dtype=np.uint8)
    keys = list(grouped_counts)
    g_off = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint32)
    g_ids = np.asarray([i for k in keys for i in k] or [0], dtype=np.int32)
    g_cnt = np.asarray([grouped_counts[k] for k in keys] or [0], dtype=np.uint32)
    args = (n, seqs, _p(pb_off, C.c_uint32), _p(pb, C.c_uint32), _p(hap, C.c_int32), _p(call, C.c_uint8), len(keys),
            _p(g_off, C.c_uint32), _p(g_ids, C.c_int32), _p(g_cnt, C.c_uint32), ploidy, mean_cov, var_cov, mean_pb_error)
    size = check(lib.gmx_infer_model(*args, None, 0))
    buf = C.create_string_buffer(size + 1)
    check(lib.gmx_infer_model(*args, buf, size + 1))
    return _json.loads(buf.value.decode())


DBG_INTERNALS, DBG_DIPLOID, DBG_NONCREDIBLE, DBG_PERMUTATIONS, DBG_RESCALE, DBG_CALL = range(6)


def genotyping_model_debug(op, alleles=(), grouped_counts=None, ploidy=1, mean_cov=10, var_cov=0, mean_pb_error=0.01, ids=(),
                           likelihoods=()) -> dict:
    """The likelihood model's pieces one by one (gmx_infer_debug; ops DBG_*): alleles as in :func:`genotyping_model`;
    ``grouped_counts`` {(ids...): count} — for DBG_CALL a list of per-haplogroup coverages; ``likelihoods`` [(value, (gt...))]."""
    import json as _json
    lib = _lib.load()
    n = len(alleles)
    seqs = (C.c_char_p * max(n, 1))(*[a[0].encode() for a in alleles])
    pb_off = np.concatenate([[0], np.cumsum([len(a[1]) for a in alleles])]).astype(np.uint32)
    pb = np.asarray([c for a in alleles for c in a[1]] or [0], dtype=np.uint32)
    hap = np.asarray([a[2] for a in alleles] or [0], dtype=np.int32)
    call = np.asarray([1 if (len(a) < 4 or a[3]) else 0 for a in alleles] or [1], dtype=np.uint8)
    if op == DBG_CALL:
        covs = list(grouped_counts or [])
        n_groups, g_off, g_ids = len(covs), np.zeros(1, np.uint32), np.zeros(1, np.int32)
        g_cnt = np.asarray(covs or [0], dtype=np.uint32)
    else:
        keys = list(grouped_counts or {})
        n_groups = len(keys)
        g_off = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint32)
        g_ids = np.asarray([i for k in keys for i in k] or [0], dtype=np.int32)
        g_cnt = np.asarray([grouped_counts[k] for k in keys] or [0], dtype=np.uint32)
    idv = np.asarray(list(ids) or [0], dtype=np.int32)
    lv = np.asarray([l[0] for l in likelihoods] or [0.0], dtype=np.float64)
    l_off = np.concatenate([[0], np.cumsum([len(l[1]) for l in likelihoods])]).astype(np.uint32)
    l_gt = np.asarray([g for l in likelihoods for g in l[1]] or [0], dtype=np.int32)
    args = (op, n, seqs, _p(pb_off, C.c_uint32), _p(pb, C.c_uint32), _p(hap, C.c_int32), _p(call, C.c_uint8), n_groups,
            _p(g_off, C.c_uint32), _p(g_ids, C.c_int32), _p(g_cnt, C.c_uint32), ploidy, mean_cov, var_cov, mean_pb_error,
            _p(idv, C.c_int32), len(ids), _p(lv, C.c_double), _p(l_off, C.c_uint32), _p(l_gt, C.c_int32), len(likelihoods))
    size = check(lib.gmx_infer_debug(*args, None, 0))
    buf = C.create_string_buffer(size + 1)
    check(lib.gmx_infer_debug(*args, buf, size + 1))
    return _json.loads(buf.value.decode())


def quasimap_reads(index: Index, read_files, seed: int, device: int = 0, rng_mode: int = RNG_LEMIRE) -> Coverage:
    """quasimap_reads (quasimap.cpp:16-57) over already-parsed reads.

    ``read_files``: list (one entry per reads file) of lists of read strings. Returns the Coverage, whose
    ``stats`` member is the reference's QuasimapReadsStats."""
    qm = Quasimapper(index, device=device, rng_mode=rng_mode)
    seeds = master_seeds(seed, [len(f) for f in read_files])
    enc = [encode_dna_bases(r) for f in read_files for r in f]
    # an unencodable read stays in the batch as an empty read: it consumes its seed and is counted as skipped
    offs = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.uint64)
    flat = np.concatenate(enc) if enc else np.zeros(0, dtype=np.uint8)
    qm.map_reads(flat, offs, seeds)
    cov = qm.coverage()
    qm.close()
    return cov
